"""kernel timeline of the LAST EncodeBlock and the LAST DecodeBlock of a rocprofv3 --kernel-trace run of tools/blockrate.py:
start, duration, gap, grid"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
DEC = ('k_synth', 'k_ms_to_lr', 'k_rice_decode', 'k_narrow')
def show(title, blk):
    print(title)
    t0 = int(blk[0]['Start_Timestamp'])
    tot, prev_end = 0, t0
    for r in blk:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        tot += e - s
        print(f"{r['Kernel_Name'][:60]:60s} +{(s - t0) / 1e3:8.1f} us dur {(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:6.1f} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
        prev_end = max(prev_end, e)
    print("kernels", len(blk), "sum dur", tot / 1e3, "span", (prev_end - t0) / 1e3)
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_prep')]
if idx:
    blk = []
    for r in rows[idx[-1]:]:
        if r['Kernel_Name'].startswith(DEC): break
        blk.append(r)
    show("--- last EncodeBlock", blk)
didx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(DEC)]
if didx:
    j = didx[-1]
    while j > 0 and rows[j - 1]['Kernel_Name'].startswith(DEC) and not rows[j]['Kernel_Name'].startswith(('k_synthesize', 'k_rice_decode')): j -= 1
    show("--- last DecodeBlock", rows[j:didx[-1] + 1])
