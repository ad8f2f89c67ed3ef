"""small driver for profiling: encode (and optionally decode) a batch of synthetic frames (stereo 16-bit unless --channels / --bits say otherwise) once or twice"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import linne_amd
from bench import synth_track, frames_from_track, KERNEL_KINDS
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--decode", action="store_true")
ap.add_argument("--preset", type=int, default=7)
ap.add_argument("--no-timing", action="store_true")
ap.add_argument("--tail", type=int, default=0, help="length of a ragged last frame (0: all frames full)")
ap.add_argument("--channels", type=int, default=2)
ap.add_argument("--bits", type=int, default=16)
a = ap.parse_args()
dev = torch.device("cuda", 0)
track = synth_track(a.frames * 10240 - ((10240 - a.tail) if a.tail else 0), a.channels, a.bits, 1, dev)
frames, nsm = frames_from_track(track, 10240)
ctx = linne_amd.Context(0, scratch_bytes=12 << 30)
shape = ctx.shape(a.channels, a.bits, 10240, a.preset, True)
ctx.enable_timing(not a.no_timing)
for r in range(a.reps):
    t0 = time.perf_counter()
    res, prm, st = ctx.encode_frames(shape, frames, nsm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {r}: encode {a.frames} frames {dt*1e3:.2f} ms -> {a.frames/dt:.0f} frames/s;", {KERNEL_KINDS[k]: round(ctx.last_ms(k), 3) for k in list(range(1, 11)) + [13, 14, 15, 16, 18, 19, 20, 21, 22, 23, 25] if ctx.last_ms(k) > 0})
if a.decode:
    for r in range(a.reps):
        w = res.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.decode_frames(shape, w, prm, nsm); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"rep {r}: decode {dt*1e3:.2f} ms -> {a.frames/dt:.0f} frames/s ok={bool(torch.equal(w, frames))}")
