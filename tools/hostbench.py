"""CPU-only timing of the host entropy stage (LINNEAmd_PackFrames) on hot-path results computed by the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import linne_amd
from refs import Oracle
from signals import music
from test_host_entropy_cpu import taps_to_arrays

F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nch, bits, block, preset = 2, 16, 10240, 7
oracle = Oracle()
x = music(nch, F * block, bits, seed=1)
enc = oracle.encoder(nch, bits, 44100, block, preset, True)
pcm = np.zeros((F, nch, block), np.int32); prm = np.zeros((F, nch, linne_amd.PARAM_WORDS), np.int32)
st = np.zeros((F, nch, linne_amd.STAT_WORDS)); res = np.zeros((F, nch, block), np.int32)
want = []
for f in range(F):
    seg = x[:, f * block:(f + 1) * block]
    pcm[f] = seg
    blk, tap, r = enc.encode_block(seg)
    want.append(blk)
    prm[f], st[f], res[f] = taps_to_arrays(tap, r, nch, preset, block)
nsm = np.full(F, block, np.uint32)
shape = linne_amd.Shape(nch, bits, block, preset, 1)
for threads in (1, 8):
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        blocks, _ = linne_amd.pack_frames(shape, pcm, res, prm, st, nsm, 0.0, threads)
        best = min(best, time.perf_counter() - t0)
    ok = all(blocks[f] == want[f] for f in range(F))
    print(f"threads {threads}: {best*1e3:.2f} ms for {F} frames -> {F/best:.0f} frames/s ({best/F/nch*1e6:.0f} us per channel-frame) ok={ok}")
import ctypes as C
lib = linne_amd.lib
cap = pcm.size * 8 + 64 * F + 64
out = np.zeros(cap, np.uint8); sizes = np.zeros(F, np.uint32)
for threads in (1, 2, 4, 8, 16):
    best = 1e9
    for rep in range(5):
        stt = C.c_double(0.0)
        t0 = time.perf_counter()
        lib.LINNEAmd_PackFrames(C.byref(shape), pcm.ctypes.data, nsm.ctypes.data, F, res.ctypes.data, prm.ctypes.data, st.ctypes.data, out.ctypes.data, cap, sizes.ctypes.data, C.byref(stt), threads)
        best = min(best, time.perf_counter() - t0)
    print(f"C call threads {threads}: {best*1e3:.2f} ms -> {F/best:.0f} frames/s")
