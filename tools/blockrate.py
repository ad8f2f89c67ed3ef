"""LINNEEncoder_EncodeBlock / LINNEDecoder_DecodeBlock called block by block (what the reference's tools/linne_codec does for
encoding, linne_codec.c:133-161): calls per second and the kernel time per call."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import linne_amd
from linne_amd.api import LinneApi
from bench import synth_track
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 200
block, nch, bits, preset = 10240, 2, 16, 7
x = np.ascontiguousarray(synth_track(nf * block, nch, bits, 3, torch.device("cuda", 0)).cpu().numpy(), dtype=np.int32)
frames = np.ascontiguousarray(x.reshape(nch, nf, block).transpose(1, 0, 2))
api = LinneApi(linne_amd.LIB_PATH)
enc = api.new_encoder(nch, bits, 44100, block, preset, True)
out = np.zeros(nch * block * 8 + 65536, dtype=np.uint8)
osz = C.c_uint32(0)
sizes = []
for rep in range(2):
    t0 = time.perf_counter()
    for f in range(nf):
        ptrs = (C.POINTER(C.c_int32) * nch)(*[frames[f, ch].ctypes.data_as(C.POINTER(C.c_int32)) for ch in range(nch)])
        assert api.L.LINNEEncoder_EncodeBlock(enc, ptrs, block, out.ctypes.data, out.size, C.byref(osz)) == 0
    dt = time.perf_counter() - t0
    print(f"EncodeBlock rep {rep}: {nf / dt:.1f} blocks/s, {dt / nf * 1e3:.3f} ms per call", flush=True)
api.L.LINNEEncoder_Destroy(enc)
