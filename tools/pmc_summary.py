"""per-dispatch PMC summary from a rocprofv3 --pmc counter_collection CSV (arg: path [substring filter])"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else "k_"
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if flt not in k or "at::" in k:
        continue
    key = (int(r["Dispatch_Id"]), k[:48])
    d = agg.setdefault(key, {"_t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "_grid": r["Grid_Size"], "_wg": r["Workgroup_Size"]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
for key, v in agg.items():
    print(key[0], key[1], "%.3f ms" % v["_t"], "grid", v["_grid"], "wg", v["_wg"], {a: int(b) for a, b in v.items() if not a.startswith("_")})
