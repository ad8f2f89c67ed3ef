"""world_size-2 `gloo` test of the N > 1 path (CPU): round-robin frame sharding, per-rank host serialisation of the
analysed frames, gather to rank 0 in stream order, barrier + MAX-over-ranks timing.  The hot-path results come from
the oracle here (no GPU); on the GPU box the same plumbing carries the HIP results (bench.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import linne_amd
    from linne_amd import sharding
    from refs import Oracle
    from signals import music_frames
    from test_host_entropy_cpu import taps_to_arrays
    nch, bits, block, preset, F = 2, 16, 1024, 0, 8
    frames = music_frames(F, nch, block, bits, seed=42)
    oracle = Oracle()
    mine = sharding.shard_round_robin(F, rank, world)
    shape = linne_amd.Shape(nch, bits, block, preset, 1)

    def analyse_and_pack():
        blocks = []
        for f in mine:
            enc = oracle.encoder(nch, bits, 44100, block, preset, True)
            tap, res = enc.hotpath(frames[f])
            enc.close()
            prm, st, full = taps_to_arrays(tap, res, nch, preset, block)
            b, _ = linne_amd.pack_frames(shape, frames[f][None], full[None], prm[None], st[None], None, 0.0, 1)
            blocks.append(b[0])
        return blocks

    holder = {}
    dt = sharding.timed_steps(lambda: holder.__setitem__("b", analyse_and_pack()), 1, dist)
    stream = sharding.gather_stream(holder["b"], F, dist)
    if rank == 0:
        single = []
        enc = oracle.encoder(nch, bits, 44100, block, preset, True)
        for f in range(F):
            blk, _, _ = enc.encode_block(frames[f])
            single.append(blk)
        enc.close()
        q.put((stream == single, dt > 0, sorted(sum([sharding.shard_round_robin(F, r, world) for r in range(world)], [])) == list(range(F))))
    dist.destroy_process_group()


def test_two_rank_sharding_reassembles_the_stream():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, timed, partition = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert same and timed and partition
