"""hunt for reads of uninitialised device memory: before every round the GPU's free memory is filled with a bit pattern (NaN doubles /
0xFF bytes / a large int) and given back, then a FRESH context encodes and decodes a batch of short ragged frames; any dependence of the
result on what the arena held -- or, with the context's streams landing on other hardware queues every round, on a missing dependence
between two of them -- shows up as a difference between rounds.  usage: poison_repro.py [rounds] [preset] [channels]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import linne_amd
from signals import music
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
preset = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
bits, block = 16, 1024
F = 20480 // nch + 77
base = np.stack([music(nch, block, bits, seed=100 * preset + k) for k in range(64)])
frames = np.ascontiguousarray(np.tile(base, ((F + 63) // 64, 1, 1))[:F])
rng = np.random.default_rng(5 + preset)
ns = np.full(F, block, dtype=np.uint32)
pool = np.array([1, 17, 130, 500, 777, block - 3, block // 2 + 1], dtype=np.uint32)
where = rng.choice(F, size=200, replace=False)
ns[where] = rng.choice(pool, size=200)
ns[-1] = 777
for f in np.flatnonzero(ns < block):
    frames[f, :, int(ns[f]):] = 0
patterns = [0xFF, 0x7F, 0x00, 0x80, 0x01, 0xAA]
first = None
bad = 0
keep_streams = []
for r in range(rounds):
    pat = patterns[r % len(patterns)]
    if os.environ.get("POISON", "1") != "0":
        junk = [torch.full((1 << 30,), pat, dtype=torch.uint8, device="cuda") for _ in range(24)]      # 24 GiB of the pattern
        torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
    extra = [torch.cuda.Stream() for _ in range(r % 9)] if os.environ.get("SHIFT_QUEUES", "1") != "0" else []      # (a HIP stream gets its hardware queue round-robin over all the process created: shift the context's)
    keep_streams.extend(extra)
    c = linne_amd.Context(0)
    try:
        shape = c.shape(nch, bits, block, preset, nch >= 2)
        res, prm, st = c.encode_frames_host(shape, frames, ns)
        marked = res.copy()
        for f in np.flatnonzero(ns < block):
            marked[f, :, int(ns[f]):] = -123456
        dec = c.decode_frames_host(shape, marked, prm, ns)
    finally:
        c.close()
    ok_dec = all(np.array_equal(dec[f, :, :int(ns[f])], frames[f, :, :int(ns[f])]) for f in range(F))
    if first is None: first = (res.copy(), prm.copy(), st.copy())
    same = np.array_equal(res, first[0]) and np.array_equal(prm, first[1]) and np.array_equal(st, first[2], equal_nan=True)
    if not (ok_dec and same):
        bad += 1
        badf = [f for f in range(F) if not np.array_equal(dec[f, :, :int(ns[f])], frames[f, :, :int(ns[f])])]
        dres = [f for f in range(F) if not np.array_equal(res[f], first[0][f])]
        dprm = [f for f in range(F) if not np.array_equal(prm[f], first[1][f])]
        print(f"round {r} pattern {pat:#x}: decode ok {ok_dec}, encode equals round 0 {same}; frames with wrong PCM {badf[:8]} (n {[int(ns[f]) for f in badf[:8]]}), residual differs {dres[:8]}, params differ {dprm[:8]}", flush=True)
    elif r % 20 == 0:
        print(f"round {r} pattern {pat:#x}: ok", flush=True)
print("rounds with a difference:", bad)
