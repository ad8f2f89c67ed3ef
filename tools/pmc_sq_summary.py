"""per-kernel summary of tools/pmc_sq.sh's two SQ counter passes -> profiles/<name>.json (arg: outdir [json path])"""
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for p in ("a", "b"):
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not (k.startswith("void k_") or k.startswith("k_")):
                continue
            d = agg.setdefault(k.replace("void ", "")[:60], collections.defaultdict(float))
            d[r["Counter_Name"]] += float(r["Counter_Value"])
            d["_n_" + p] += 1.0 / 8
res = {}
for k, d in agg.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    e = {"dispatches": round(d["_n_a"]), "wave_cycles": wc,
         "active_inst_any_frac": d["SQ_ACTIVE_INST_ANY"] / wc, "wait_inst_any_frac": d["SQ_WAIT_INST_ANY"] / wc, "wait_any_frac": d["SQ_WAIT_ANY"] / wc,
         "active_inst_valu_frac": d["SQ_ACTIVE_INST_VALU"] / wc, "active_inst_lds_frac": d["SQ_ACTIVE_INST_LDS"] / wc, "wait_inst_lds_frac": d["SQ_WAIT_INST_LDS"] / wc,
         "busy_cycles": d["SQ_BUSY_CYCLES"],
         "insts_valu": d["SQ_INSTS_VALU"], "insts_lds": d["SQ_INSTS_LDS"], "insts_salu": d["SQ_INSTS_SALU"], "insts_smem": d["SQ_INSTS_SMEM"],
         "lds_bank_conflict_over_idx_active": (d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]) if d["SQ_LDS_IDX_ACTIVE"] else None,
         "lds_idx_active": d["SQ_LDS_IDX_ACTIVE"], "waves": d["SQ_WAVES"]}
    res[k] = e
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["wave_cycles"]))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
for k, e in list(res.items())[:12]:
    print(f"{k:58s} waveCyc {e['wave_cycles']:.3g} act {e['active_inst_any_frac']:.2f} (valu {e['active_inst_valu_frac']:.2f} lds {e['active_inst_lds_frac']:.2f}) waitInst {e['wait_inst_any_frac']:.2f} (lds {e['wait_inst_lds_frac']:.2f}) waitAny {e['wait_any_frac']:.2f} conflicts/ldsActive {e['lds_bank_conflict_over_idx_active']}")
