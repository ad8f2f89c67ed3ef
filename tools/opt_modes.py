"""Throughput of the optional analysis modes -- `-a N` (auxiliary-function iterations) and `-l` (the network trainer) -- through
the drop-in API on the GPU, beside the reference library (oracle/_ref) on the host cores.  Usage:
    python tools/opt_modes.py [minutes_gpu=2] [frames_per_cpu_thread=1]
Prints one JSON line per mode."""
import ctypes as C, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import linne_amd
from linne_amd.api import LinneApi
from bench import synth_track, host_cores, _checkers

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
block, nch, bits, preset = 10240, 2, 16, 7
ns = int(minutes * 60 * 44100)
x = np.ascontiguousarray(synth_track(ns, nch, bits, 3, torch.device("cuda", 0)).cpu().numpy(), dtype=np.int32)
nf = (ns + block - 1) // block
prod = LinneApi(linne_amd.LIB_PATH)
_, REF_SO, _, reference_available = _checkers()
ref = LinneApi(REF_SO) if reference_available() else None
cores = host_cores()


def cpu_rate(af, learn):
    frames = np.ascontiguousarray(x[:, :cores * per * block].reshape(nch, cores * per, block).transpose(1, 0, 2))

    def work(t):
        enc = ref.new_encoder(nch, bits, 44100, block, preset, True, af_iters=af, learning=learn)
        out = np.zeros(nch * block * 8 + 65536, dtype=np.uint8)
        osz = C.c_uint32(0)
        for f in range(t * per, (t + 1) * per):
            ptrs = (C.POINTER(C.c_int32) * nch)(*[frames[f, ch].ctypes.data_as(C.POINTER(C.c_int32)) for ch in range(nch)])
            assert ref.L.LINNEEncoder_EncodeBlock(enc, ptrs, block, out.ctypes.data, out.size, C.byref(osz)) == 0
        ref.L.LINNEEncoder_Destroy(enc)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return cores * per / (time.perf_counter() - t0)


for name, af, learn, frac in (("-a 0", 0, 0, 1.0), ("-a 1", 1, 0, 1.0), ("-a 3", 3, 0, 1.0), ("-l", 0, 1, 0.25)):
    n = int(ns * frac)
    xs = np.ascontiguousarray(x[:, :n])
    enc = prod.new_encoder(nch, bits, 44100, block, preset, True, af_iters=af, learning=learn)
    ptrs = (C.POINTER(C.c_int32) * nch)(*[xs[ch].ctypes.data_as(C.POINTER(C.c_int32)) for ch in range(nch)])
    buf = np.zeros(xs.size * 4 + 65536, dtype=np.uint8)
    osz = C.c_uint32(0)
    assert prod.L.LINNEEncoder_EncodeWhole(enc, ptrs, min(n, 8 * block), buf.ctypes.data, buf.size, C.byref(osz)) == 0      # warm-up: context, slots
    t0 = time.perf_counter()
    assert prod.L.LINNEEncoder_EncodeWhole(enc, ptrs, n, buf.ctypes.data, buf.size, C.byref(osz)) == 0
    dt = time.perf_counter() - t0
    prod.L.LINNEEncoder_Destroy(enc)
    out = buf[:osz.value]
    frames = (n + block - 1) // block
    line = {"mode": name, "gpu_frames_per_s": frames / dt, "gpu_frames": frames, "bytes": len(out)}
    if ref is not None:
        line["cpu_reference_frames_per_s"] = cpu_rate(af, learn)
        line["cpu_cores"] = cores
        line["speedup"] = line["gpu_frames_per_s"] / line["cpu_reference_frames_per_s"]
    print(json.dumps(line), flush=True)
