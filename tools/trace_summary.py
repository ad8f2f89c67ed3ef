"""prints per-dispatch durations of our kernels from a rocprofv3 --kernel-trace CSV (arg: path)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Kernel_Name"]
    if "k_" in n and "at::" not in n:
        print(f'{n[:60]:60s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:8.3f} ms  grid {r.get("Grid_Size_X", r.get("Grid_Size"))} wg {r.get("Workgroup_Size_X", r.get("Workgroup_Size"))} vgpr {r.get("VGPR_Count")} lds {r.get("LDS_Block_Size")}')
