"""Certificate telemetry: over many random configurations, encode every batch twice -- with the certified
order-free unit-count search (default) and with LINNE_AMD_EXACT=1 (every search through the ordered unfused chains) -- and check
that the bytes agree; log the smallest certified margin (LINNEAmd_GetLastMinMargin: gap between the winner's upper bound and the
runner-up's lower bound, relative to the winning mean) and how many searches took the exact fallback by themselves.
usage: python tools/margin_sweep.py [seeds=400] [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import linne_amd
from signals import music, waveform

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
out = sys.argv[2] if len(sys.argv) > 2 else None
os.environ.pop("LINNE_AMD_EXACT", None)
cert = linne_amd.Context(0, use_torch_stream=False)
os.environ["LINNE_AMD_EXACT"] = "1"
exact = linne_amd.Context(0, use_torch_stream=False)
os.environ.pop("LINNE_AMD_EXACT")
res = {"seeds": nseeds, "mismatches": [], "min_margin_music": None, "min_margin_all": None, "searches": 0, "own_fallbacks": 0, "by_kind": {}}
t0 = time.time()
for seed in range(nseeds):
    rng = np.random.default_rng(5000 + seed)
    nch = int(rng.integers(1, 5)); bits = int(rng.choice([8, 16, 24])); preset = int(rng.integers(0, 8))
    block = int(rng.choice([1024, 2048, 4096, 10240, 2 * int(rng.integers(200, 1500))]))
    maxp = 32 if preset < 2 else (64 if preset < 5 else 128)
    if block <= maxp:
        block = 1024
    F = int(rng.integers(2, 7))
    kind = ["music", "music", "music", "chirp", "sine", "gauss_noise", "mixed"][int(rng.integers(0, 7))]
    if kind == "music":
        x = music(nch, F * block, bits, seed=int(rng.integers(1 << 30)))
    elif kind == "mixed":
        x = music(nch, F * block, bits, seed=int(rng.integers(1 << 30))); x[:, block:2 * block] = 0; x[:, 2 * block:3 * block] //= 128
    else:
        x = waveform(kind, nch, F * block, bits, seed=int(rng.integers(1 << 30)))
    frames = np.ascontiguousarray(x.reshape(nch, F, block).transpose(1, 0, 2))
    ns = np.full(F, block, dtype=np.uint32); ns[-1] = int(rng.integers(max(maxp + 1, 130), block + 1)); frames[-1, :, ns[-1]:] = 0
    shape = cert.shape(nch, bits, block, preset, nch >= 2 and bool(rng.integers(0, 2)))
    try:
        a = cert.encode_frames_host(shape, frames, ns)
    except linne_amd.LinneAmdError:
        continue                                 # odd analysis length etc.: not this sweep's subject
    fb, mg = cert.last_fallback_count(), cert.last_min_margin()
    b = exact.encode_frames_host(shape, frames, ns)
    same = all(np.array_equal(p, q, equal_nan=True) for p, q in zip(a, b))
    nsearch = F * nch * linne_amd.PRESET_NUM_REGULARS[preset] * len(linne_amd.PRESET_LAYERS[preset])
    res["searches"] += nsearch; res["own_fallbacks"] += fb
    k = res["by_kind"].setdefault(kind, {"batches": 0, "min_margin": None, "fallbacks": 0})
    k["batches"] += 1; k["fallbacks"] += fb
    if mg < 1e300:
        k["min_margin"] = mg if k["min_margin"] is None else min(k["min_margin"], mg)
        res["min_margin_all"] = mg if res["min_margin_all"] is None else min(res["min_margin_all"], mg)
        if kind == "music":
            res["min_margin_music"] = mg if res["min_margin_music"] is None else min(res["min_margin_music"], mg)
    if not same:
        res["mismatches"].append({"seed": seed, "nch": nch, "bits": bits, "preset": preset, "block": block, "kind": kind})
res["seconds"] = time.time() - t0
res["note"] = ("margin = (lower bound of the runner-up's mean - upper bound of the winner's mean) / winner's mean over the searches the certificate decided; "
               "rel = (2 na + 8) 2^-53 ~ 2e-12 and the slack are what it has to exceed; own_fallbacks = searches the certificate handed to the exact chains by itself")
print(json.dumps(res, indent=1))
if out:
    json.dump(res, open(out, "w"), indent=1)
cert.close(); exact.close()
sys.exit(1 if res["mismatches"] else 0)
