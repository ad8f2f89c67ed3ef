"""timeline of the LAST LINNEEncoder_EncodeWhole of a `rocprofv3 --kernel-trace --memory-copy-trace` run: per queue, the busy spans of
kernels (merged) and the copies, in ms relative to the call's first k_prep (arg: the directory rocprofv3 wrote)"""
import csv, glob, os, sys, collections
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
ev = []
for r in csv.DictReader(open(kt[0])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:34], "q" + r.get("Queue_Id", "?")))
if mt:
    for r in csv.DictReader(open(mt[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?")[12:], "copy"))
ev.sort()
emit = [e for e in ev if e[2].startswith("k_rice_emit")]
if not emit:
    sys.exit("no k_rice_emit in the trace")
# the last EncodeWhole: walk back from the last k_rice_emit to the k_prep that starts its call (gap of > 30 ms in front)
last = emit[-1][0]
preps = [e for e in ev if e[2].startswith("k_prep") and e[0] <= last]
t0 = preps[-1][0]
for a, b in zip(reversed(preps[:-1]), reversed(preps[1:])):
    if b[0] - a[0] > 60e6: break
    t0 = a[0]
per = collections.defaultdict(list)
for s, e, n, q in ev:
    if s >= t0 - 5e6 and s <= last + 30e6:
        per[q].append((s, e, n))
for q, lst in sorted(per.items()):
    busy = sum(e - s for s, e, _ in lst) / 1e6
    names = collections.Counter(n for _, _, n in lst)
    print(f"{q:6s} busy {busy:8.2f} ms, first {(lst[0][0]-t0)/1e6:8.2f}, last end {(max(e for _, e, _ in lst)-t0)/1e6:8.2f}: " + ", ".join(f"{n} x{c}" for n, c in names.most_common(6)))
print("k_search_long spans (ms from t0):", [(round((s - t0) / 1e6, 1), round((e - s) / 1e6, 1), q) for s, e, n, q in ev if n.startswith("k_search_long") and s >= t0 and s <= last])
