"""decode time of DecodeFramesDevice against the batch size, under each form of the synthesis (where the default's
threshold between the latency form and the throughput form comes from): python tools/decode_crossover.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import linne_amd
from bench import synth_track
dev = torch.device("cuda", 0)
S, C = 10240, 2
for F in (256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288):
    x = synth_track(F * S, C, 16, 5, dev).reshape(C, F, S).permute(1, 0, 2).contiguous()
    line = [f"frames {F:5d} (channel-frames {F * C:5d}):"]
    for form in ("pipe", "rows", "rows4", "rows_nf", "lanes"):        # rows_nf: LINNE_AMD_DECODE_FUSED=0 (layer 0, de-emphasis, MS -> LR as launches of their own)
        if form == "pipe" and F > 4096:
            continue
        os.environ["LINNE_AMD_DECODE_KERNEL"] = "rows" if form in ("rows4", "rows_nf") else form
        os.environ["LINNE_AMD_DECODE_ROWS8"] = "0" if form == "rows4" else "1"
        os.environ["LINNE_AMD_DECODE_FUSED"] = "0" if form == "rows_nf" else "1"
        ctx = linne_amd.Context(0)
        shape = ctx.shape(C, 16, S, 7, True)
        res, prm, st = ctx.encode_frames(shape, x)
        work = res.clone()
        ctx.decode_frames(shape, work, prm); ctx.synchronize()
        ok = bool(torch.equal(work, x))
        ts = []
        for _ in range(5):
            work.copy_(res); torch.cuda.synchronize()
            t0 = time.perf_counter(); ctx.decode_frames(shape, work, prm); ctx.synchronize(); ts.append(time.perf_counter() - t0)
        line.append(f"{form} {min(ts) * 1e3:6.3f} ms{'' if ok else ' (MISMATCH)'}")
        ctx.close()
    print("  ".join(line), flush=True)
