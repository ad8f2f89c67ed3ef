"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md
prescribes) of tools/kbench.py --frames F: writes profiles/pmc_latest.json.  Counters are KiB; FETCH_SIZE is doubled
(gfx950 reports half of the bytes of wide coalesced reads); WRITE_SIZE is used as is.
usage: pmc_traffic.py fetch.csv write.csv frames channels [note, e.g. the commit the library was built from]"""
import csv, json, sys, collections
fetch_csv, write_csv, frames, channels = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
KIND = [("k_search_long", "k_search_long<P> (search of the long layer over the shared window + fused one-unit forward)"),
        ("k_fir2<2, false, true>", "k_fir2<2,false,true> (search of the long layer + fused one-unit forward; frames k_search_long does not take)"),
        ("k_fir_small<16, false", "k_fir_small<P,false,*> (search of the last, short layer)"), ("k_fir_small<8, false", "k_fir_small<P,false,*> (search of the last, short layer)"),
        ("k_fir_small<4, true", "k_fir_small<P,true,*> (search of layer 0 + fused one-unit forward)"), ("k_fir_small<2, true", "k_fir_small<P,true,*> (search of layer 0 + fused one-unit forward)"),
        ("k_fir2<1, false, true>", "k_fir2<1,false,true> (forward, jobs with several units)"), ("k_fir2<1, false, false>", "k_fir2<1,false,false> (forward of the last layer, frames k_fwd_loss does not take)"), ("k_last_layer", "k_last_layer<P> / k_fwd_loss<P> (last layer: exact search + forward pass + ordered loss in one launch / forward pass + loss)"), ("k_fwd_loss", "k_last_layer<P> / k_fwd_loss<P> (last layer: exact search + forward pass + ordered loss in one launch / forward pass + loss)"),
        ("k_fir2<1, true", "k_fir2<1,true,*> (forward of layer 0, jobs with several units)"),
        ("k_autocorr_hist<128, 0>", "k_autocorr_hist<P,0> (long layer, one-unit trial)"), ("k_autocorr_hist<64, 0>", "k_autocorr_hist<P,0> (long layer, one-unit trial)"), ("k_autocorr_hist<128, 1>", "k_autocorr_hist<P,1> (long layer, two-unit trial)"), ("k_autocorr_sub", "k_autocorr_sub<P> (long layer, trials of order <= 32)"), ("k_autocorr2", "k_autocorr2"), ("k_autocorr_lane", "k_autocorr_lane"), ("k_levinson", "k_levinson_lds"), ("k_prep_slow", "k_prep_slow (the ordered pre-emphasis chains of loud 24-bit material; 16-bit: its blocks leave at once)"), ("k_prep", "k_prep"), ("k_finalize", "k_finalize"), ("k_quantize", "k_finalize"), ("k_fir_cascade", "k_finalize"),
        ("k_synth_l0_de", "k_synth_l0_de (layer 0 + de-emphasis + MS -> LR in one launch, tiles in LDS)"),
        ("k_synth_rows8", "k_synth_rows8<PB> / k_synth_rows<0> (a short layer: eight / four channel-frames per wave)"), ("k_synth_rows<0", "k_synth_rows8<PB> / k_synth_rows<0> (a short layer: eight / four channel-frames per wave)"),
        ("k_synth_rows", "k_synth_rows<NCH> (a long layer: four channel-frames per wave, the old taps on the matrix unit)"), ("k_deemph_lr", "k_deemph_lr (de-emphasis behind layer 0, MS -> LR on the way out; LINNE_AMD_DECODE_FUSED=0)"),
        ("k_synth_big", "k_synth_big<P> (synthesis of the long layer)"), ("k_synth_small", "k_synth_small<P> (synthesis of the short layers, de-emphasis)"),
        ("k_synthesize", "k_synthesize (one wave per channel-frame, all layers)"), ("k_ms_to_lr", "k_ms_to_lr"), ("k_chain_sum", "k_chain_sum<1>"), ("k_stats", "k_stats"), ("k_rice_plan", "k_rice_plan"), ("k_rice_emit", "k_rice_emit")]
def load(path, counter):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return per
f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
out = {"_source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on tools/kbench.py --frames {frames} "
       f"({frames * channels} channel-frames); counters are KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; "
       "per kernel kind: all launches of one encode call summed, then divided by the launches and by the channel-frames of a launch"}
def collect(per):
    agg = collections.defaultdict(list)
    for _, (name, val) in sorted(per.items()):
        for sub, kind in KIND:
            if sub in name:
                agg[kind].append(val); break
    return agg
fa, wa = collect(f), collect(w)
for kind in fa:
    fk, wk = fa[kind], wa.get(kind, [0.0] * len(fa[kind]))
    n = len(fk)
    bytes_total = (2.0 * sum(fk) + sum(wk)) * 1024.0
    out[kind] = {"launches_sampled": n, "fetch_kib_raw": [int(v) for v in fk], "write_kib": [int(v) for v in wk],
                 "hbm_bytes_per_channel_frame_per_launch": int(bytes_total / n / (frames * channels))}
# the whole step: every kernel of the encode call (decode call) summed, per channel-frame -- against the algorithmic 82 552 B
DECODE = ("k_synth", "k_ms_to_lr", "k_deemph_lr")
tot = {"encode": 0.0, "decode": 0.0}
for kind, v in out.items():
    if kind.startswith("_") or kind.startswith("k_rice"): continue
    side = "decode" if kind.startswith(DECODE) else "encode"
    tot[side] += (2.0 * sum(v["fetch_kib_raw"]) + sum(v["write_kib"])) * 1024.0
out["_whole_step"] = {"encode_hbm_bytes_per_channel_frame": int(tot["encode"] / (frames * channels)), "decode_hbm_bytes_per_channel_frame": int(tot["decode"] / (frames * channels)),
                      "algorithmic_bytes_per_channel_frame": 82552, "encode_over_algorithmic": round(tot["encode"] / (frames * channels) / 82552.0, 2),
                      "decode_over_algorithmic": round(tot["decode"] / (frames * channels) / 82552.0, 2)}
# the commit the library was built from: the GPU box has no .git -- the caller passes it (tools/round_profiles.sh takes it from the
# build container: `bash tools/round_profiles.sh r04 "$(git rev-parse --short HEAD)"`)
commit = ""
try:
    import subprocess
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    pass
note = sys.argv[5] if len(sys.argv) > 5 else ""
out["_source"] += "; tree " + (commit or note or "UNKNOWN (no commit id was passed)")
if commit and note: out["_source"] += "; " + note
json.dump(out, open("profiles/pmc_latest.json", "w"), indent=1)
for k, v in out.items():
    if not k.startswith("_"): print(k, v["launches_sampled"], v["hbm_bytes_per_channel_frame_per_launch"])
