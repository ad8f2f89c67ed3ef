"""end-to-end timing of the drop-in API (host planes in, .lnn bytes out and back): LINNEEncoder_EncodeWhole /
LINNEDecoder_DecodeWhole on one handle each, caller buffers touched beforehand; rep 0 pays the one-time set-up of the
handle's GPU context and pinned staging slots, later reps are the steady state."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import linne_amd
from refs import LinneApi, _planar_ptrs, _RefDecoderConfig
from bench import synth_track
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nch = int(sys.argv[3]) if len(sys.argv) > 3 else 2          # e.g. BASELINE configs[4]: e2e.py 10 3 8 24 96000
bits = int(sys.argv[4]) if len(sys.argv) > 4 else 16
rate = int(sys.argv[5]) if len(sys.argv) > 5 else 44100
ns = int(minutes * 60 * rate)
x = np.ascontiguousarray(synth_track(ns, nch, bits, 3, torch.device("cuda", 0), rate=float(rate)).cpu().numpy(), dtype=np.int32)
api = LinneApi(linne_amd.LIB_PATH)
L = api.L
nf = (ns + 10239) // 10240
enc = api.new_encoder(nch, bits, rate, 10240, 7, nch >= 2)
cfg = _RefDecoderConfig(nch, 5, 128, 1)
dec = L.LINNEDecoder_Create(C.byref(cfg), None, 0)
xp, _k1 = _planar_ptrs(x)
cap = x.size * 4 + 65536
out = np.ones(cap, dtype=np.uint8)
back = np.ones_like(x)
bp, _k2 = _planar_ptrs(back)
osz = C.c_uint32(0)
for rep in range(reps):
    t0 = time.perf_counter()
    r1 = L.LINNEEncoder_EncodeWhole(enc, xp, ns, out.ctypes.data, cap, C.byref(osz))
    t1 = time.perf_counter()
    r2 = L.LINNEDecoder_DecodeWhole(dec, out.ctypes.data, osz.value, bp, nch, ns)
    t2 = time.perf_counter()
    print(f"rep {rep}: {nf} frames; EncodeWhole {t1-t0:.3f} s -> {nf/(t1-t0):.0f} frames/s; DecodeWhole {t2-t1:.3f} s -> {nf/(t2-t1):.0f} frames/s; "
          f"ratio {osz.value/(x.size*(bits//8)):.3f}; ok={r1 == 0 and r2 == 0 and np.array_equal(back, x)}", flush=True)
L.LINNEEncoder_Destroy(enc); L.LINNEDecoder_Destroy(dec)
