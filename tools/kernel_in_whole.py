"""kernel time inside the LAST LINNEEncoder_EncodeWhole of a `rocprofv3 --kernel-trace` run (arg: the directory rocprofv3 wrote; optional: a
substring of the kernel names to list launch by launch): per kernel the launches and their summed duration over the 200 ms in front of
the call's last k_rice_emit"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
key = sys.argv[2] if len(sys.argv) > 2 else None
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]) for r in csv.DictReader(open(f)))
emit = [e for e in ev if e[2].startswith("k_rice_emit")]
last = emit[-1][1]; t0 = last - 200e6
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in ev:
    if s >= t0 and e <= last + 1e6:
        agg[n][0] += 1; agg[n][1] += (e - s) / 1e6
        if key and key in n: print(f"  {n} at {(s - t0) / 1e6:7.2f} ms: {(e - s) / 1e6:.3f} ms")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]: print(f"{n:42s} {v[0]:4d} {v[1]:8.2f}")
print("sum:", round(sum(v[1] for v in agg.values()), 1))
