"""How wide are the gaps between the trials of the unit-count search (libs/linne_network/src/linne_network.c:318-341), measured against
what an fp32 evaluation of the order-free trials could certify?  CPU only: the oracle (test infrastructure) encodes frames of the
bench's kind of material with a tap on every search, and for every search of the long layer this prints how many could be decided by
trial means that carry an fp32 error bound instead of the FP64 one ("The certified unit-count search").

An fp32 evaluation (inputs rounded to fp32, fp32 fused multiply-adds, products summed in any order) of x + sum_k h_k x_k over np taps
lies within  (np + 4) 2^-24 (|x| + sum |h_k x_k|)  of the exact value; the trial's mean then carries the slack
(np + 4) 2^-24 max|x| (1 + max_unit sum|h_k|).  The one-unit trial stays exact (its chain is the forward pass).
usage: python3 tools/search_margins.py [frames] [out.json]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from bindings import Oracle
from signals import music

nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 48
block, nch, bits, preset = 10240, 2, 16, 7
o = Oracle()
o.L.oracle_set_trial_tap.argtypes = [C.c_void_p, C.c_uint32]
o.L.oracle_trial_tap_count.restype = C.c_uint32
x = music(nch, nframes * block, bits, seed=1)
frames = x.reshape(nch, nframes, block).transpose(1, 0, 2)
cap = nframes * nch * 4 * 32 * 2
buf = np.zeros((cap, 5))
o.L.oracle_set_trial_tap(buf.ctypes.data, cap)
for f in range(nframes):
    enc = o.encoder(nch, bits, 44100, block, preset, True)
    enc.hotpath(np.ascontiguousarray(frames[f]))
    enc.close()
n = o.L.oracle_trial_tap_count()
o.L.oracle_set_trial_tap(None, 0)
rec = buf[:n]
# group the records into searches: a search = consecutive records of one num_params with nunits 1, 2, 4, ...
searches, cur = [], []
for r in rec:
    if r[1] == 1 and cur:
        searches.append(cur); cur = []
    cur.append(r)
if cur:
    searches.append(cur)
out = {}
for P in (128, 16, 4):
    S = [s for s in searches if s[0][0] == P]
    if not S:
        continue
    gaps, ok32, ok64, wins, okx = [], 0, 0, {}, [0, 0, 0]
    for s in S:
        means = np.array([t[2] for t in s]); units = [int(t[1]) for t in s]
        b = int(np.argmin(means))           # (strict < from the left = first minimum)
        wins[units[b]] = wins.get(units[b], 0) + 1
        others = np.delete(means, b)
        gaps.append(float((others.min() - means[b]) / means[b]) if len(others) else 1.0)
        def certified(eps):
            lo, hi = [], []
            for t in s:
                np_ = P / t[1]
                sl = 0.0 if t[1] == 1 else (np_ + 4) * eps * t[4] * (1.0 + t[3])
                lo.append(t[2] - sl); hi.append(t[2] + sl)
            return all(hi[b] < lo[k] for k in range(len(s)) if k != b)
        ok32 += certified(2.0 ** -24); ok64 += certified(2.0 ** -53)
        for kx, e in enumerate((2.0 ** -22, 2.0 ** -20, 2.0 ** -18)):
            okx[kx] += certified(e)
    g = np.array(gaps)
    out[P] = {"searches": len(S), "winner_units": wins, "relative_gap_to_runner_up": {q: float(np.quantile(g, q)) for q in (0.01, 0.05, 0.25, 0.5, 0.75)},
              "certified_with_fp32_bounds": ok32 / len(S), "certified_with_fp64_bounds": ok64 / len(S),
              "certified_with_bounds_4x_16x_64x_the_fp32_one": [v / len(S) for v in okx]}
    print(P, json.dumps(out[P]))
if len(sys.argv) > 2:
    json.dump({"material": f"{nframes} stereo frames of tests/signals.music (the bench's recipe), -m 7", "per_layer_order": out}, open(sys.argv[2], "w"), indent=1)
