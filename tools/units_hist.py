import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, linne_amd
from bench import synth_track, frames_from_track
dev = torch.device("cuda", 0)
track = synth_track(1024 * 10240, 2, 16, 1, dev)
frames, nsm = frames_from_track(track, 10240)
ctx = linne_amd.Context(0, scratch_bytes=6 << 30)
shape = ctx.shape(2, 16, 10240, 7, True)
res, prm, st = ctx.encode_frames(shape, frames, nsm)
torch.cuda.synchronize()
u = prm[:, :, linne_amd.PRM_UNITS:linne_amd.PRM_UNITS + 3].cpu().numpy().reshape(-1, 3)
for l in range(3):
    vals, cnt = np.unique(u[:, l], return_counts=True)
    print("layer", l, dict(zip(vals.tolist(), cnt.tolist())))
print("best regulariser", np.unique(st[:, :, linne_amd.ST_BEST].cpu().numpy(), return_counts=True))
