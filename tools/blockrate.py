"""LINNEEncoder_EncodeBlock / LINNEDecoder_DecodeBlock called block by block (what the reference's tools/linne_codec does for
encoding, linne_codec.c:133-161, and tools/linne_player from its audio callback for decoding, linne_player.c:66-118): ms per call
through the unchanged 13-symbol API.  The same leg bench.py puts into its line as `block_at_a_time`.
usage: python tools/blockrate.py [blocks]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 200
block, nch, bits, preset = 10240, 2, 16, 7
x = bench.synth_track(nf * block, nch, bits, 3, torch.device("cuda", 0))
frames, _ = bench.frames_from_track(x, block)
frames = frames.cpu().numpy()
for rep in range(2):
    rec = bench.block_at_a_time(frames, bits, 44100, block, preset, True, None, nblocks=nf)
    print(json.dumps({k: rec[k] for k in ("encode_ms", "decode_ms", "blocks", "decode_bit_exact")}), flush=True)
