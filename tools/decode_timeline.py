"""timeline of the LAST LINNEDecoder_DecodeWhole of a `rocprofv3 --kernel-trace --memory-copy-trace` run of tools/e2e.py: kernels and
copies with start / end in ms relative to the call's first Rice launch (arg: the directory rocprofv3 wrote, prefix p)"""
import csv, glob, os, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
ev = []
for r in csv.DictReader(open(kt[0])):
    n = r["Kernel_Name"]
    if n.startswith("k_") or "k_" in n.split("(")[0]:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][:40], r.get("Queue_Id", "?")))
if mt:
    for r in csv.DictReader(open(mt[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?"), "-"))
ev.sort()
rice = [e for e in ev if "k_rice_decode" in e[2]]
if not rice:
    sys.exit("no k_rice_decode in the trace")
# the last call: the rice launches that follow the last gap of > 50 ms
starts = [rice[0][0]]
for a, b in zip(rice, rice[1:]):
    if b[0] - a[1] > 50e6:
        starts.append(b[0])
t0 = starts[-1]
print("ms from the call's first Rice launch: start, end, duration, what, queue")
for s, e, n, q in ev:
    if s >= t0 - 20e6 and s < t0 + 200e6 and (e - s) > 20e3:
        print(f"{(s - t0) / 1e6:8.2f} {(e - t0) / 1e6:8.2f} {(e - s) / 1e6:7.2f}  {n:40s} q{q}")
