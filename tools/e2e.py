"""end-to-end timing of the drop-in API (host buffers in, .lnn bytes out): LINNEEncoder_EncodeWhole / LINNEDecoder_DecodeWhole"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import linne_amd
from refs import LinneApi
from bench import synth_track
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
ns = int(minutes * 60 * 44100)
x = synth_track(ns, 2, 16, 3, torch.device("cuda", 0)).cpu().numpy()
api = LinneApi(linne_amd.LIB_PATH)
for rep in range(2):
    t0 = time.perf_counter(); lnn = api.encode_whole(x, 16, 44100, 10240, 7, True); t1 = time.perf_counter()
    ret, dec = api.decode_whole(lnn); t2 = time.perf_counter()
    nf = (ns + 10239) // 10240
    print(f"rep {rep}: {nf} frames; EncodeWhole {t1-t0:.3f} s -> {nf/(t1-t0):.0f} frames/s; DecodeWhole {t2-t1:.3f} s -> {nf/(t2-t1):.0f} frames/s; ratio {len(lnn)/(x.size*2):.3f}; ok={ret == 0 and np.array_equal(dec, x)}")
