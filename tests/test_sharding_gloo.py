"""world_size-2 `gloo` test of the N > 1 data path (CPU): linne_amd.sharding.ChunkExchange -- the pipelined point-to-point
scatter of frame chunks from the root and gather of residual + params + stats back (on the GPU box the same code runs on
backend "nccl" = RCCL with device tensors).  The per-chunk analysis comes from the oracle here (no GPU); the root then
serialises the gathered frames IN STREAM ORDER and must get the single-stream encoder's bytes -- also around SILENT / RAW
blocks, where the block-type decision carries state from block to block (quirk Q2)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NCH, BITS, BLOCK, PRESET = 2, 16, 1024, 0


def _stream():
    from signals import music, waveform
    parts = [music(NCH, 5 * BLOCK, BITS, seed=42), np.zeros((NCH, BLOCK), dtype=np.int32), waveform("white_noise", NCH, 2 * BLOCK, BITS, seed=3),
             music(NCH, 4 * BLOCK + 300, BITS, seed=43)]
    x = np.concatenate(parts, axis=1)
    F = (x.shape[1] + BLOCK - 1) // BLOCK
    frames = np.zeros((F, NCH, BLOCK), dtype=np.int32)
    ns = np.full(F, BLOCK, dtype=np.uint32)
    for f in range(F):
        seg = x[:, f * BLOCK:(f + 1) * BLOCK]
        frames[f, :, :seg.shape[1]] = seg
        ns[f] = seg.shape[1]
    return x, frames, ns


def _worker(rank, world, port, chunk, q, own_group=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import linne_amd
    from linne_amd import sharding
    from refs import Oracle
    from test_host_entropy_cpu import taps_to_arrays
    x, frames, ns = _stream()
    F = frames.shape[0]
    oracle = Oracle()
    shape = linne_amd.Shape(NCH, BITS, BLOCK, PRESET, 1)
    seen = []

    def analyse(inputs, nsm, outputs):           # stands in for LINNEAmd_EncodeFramesDevice: chunk of PCM -> residual, params, stats
        (pcm,), (res, prm, st) = inputs, outputs
        seen.append(pcm.shape[0])
        for i in range(pcm.shape[0]):
            n = int(nsm[i])
            enc = oracle.encoder(NCH, BITS, 44100, BLOCK, PRESET, True)
            tap, r = enc.hotpath(pcm[i].numpy()[:, :n])
            enc.close()
            p, s, full = taps_to_arrays(tap, r, NCH, PRESET, BLOCK)
            res[i] = torch.from_numpy(full); prm[i] = torch.from_numpy(p); st[i] = torch.from_numpy(s)

    # own_group: the transfers run on a group of their own (bench.py keeps its control traffic on the default group and hands the
    # exchange the group RCCL serves)
    group = dist.new_group(backend="gloo") if own_group else None
    ex = sharding.ChunkExchange(dist, F, chunk, [((NCH, BLOCK), torch.int32)],
                                [((NCH, BLOCK), torch.int32), ((NCH, linne_amd.PARAM_WORDS), torch.int32), ((NCH, linne_amd.STAT_WORDS), torch.float64)],
                                torch.device("cpu"), root=0, group=group)
    if rank == 0:
        pcm = torch.from_numpy(frames)
        out = [torch.zeros((F, NCH, BLOCK), dtype=torch.int32), torch.zeros((F, NCH, linne_amd.PARAM_WORDS), dtype=torch.int32),
               torch.zeros((F, NCH, linne_amd.STAT_WORDS), dtype=torch.float64)]
    else:
        pcm, out = None, None
    holder = {}
    for step in range(2):                        # twice: the buffers and the op order must survive a second step
        dt = sharding.timed_steps(lambda: ex.run(analyse, ns, [pcm] if rank == 0 else None, out), 1, dist)
    mine = sum(1 for c in range(len(ex.chunks)) if c % world == rank)
    ok_share = (len(seen) == 2 * mine)
    if rank == 0:
        blocks, _ = linne_amd.pack_frames(shape, frames, out[0].numpy(), out[1].numpy(), out[2].numpy(), ns, 0.0, 2)
        stream = b"".join(blocks)
        want = oracle.encode_whole(x, BITS, 44100, BLOCK, PRESET, True)
        types = {b[8] for b in blocks}
        q.put(("root", stream == want[30:], dt > 0, ok_share, types == {0, 1, 2}))
    else:
        q.put(("peer", True, dt > 0, ok_share, True))
    dist.barrier()
    dist.destroy_process_group()


def _run(chunk, own_group=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + chunk
    procs = [ctx.Process(target=_worker, args=(r, 2, port, chunk, q, own_group)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for who, same, timed, share, types in got:
        assert same and timed and share and types, (who, same, timed, share, types)


def test_two_rank_scatter_gather_reassembles_the_stream():
    _run(chunk=3)           # 13 frames: chunks 3 3 3 3 1 -> three rounds, the last one with the root only


def test_two_rank_scatter_gather_on_a_group_of_its_own():
    _run(chunk=4, own_group=True)


def test_two_rank_scatter_gather_with_one_chunk_per_rank():
    _run(chunk=7)           # chunks 7 6: a single round, nothing to prefetch


def test_chunk_exchange_without_a_process_group():
    """world size 1 (no torch.distributed at all): the root just walks its chunks"""
    sys.path.insert(0, ROOT)
    from linne_amd import sharding
    ex = sharding.ChunkExchange(None, 10, 4, [((2,), torch.int32)], [((2,), torch.int32)], torch.device("cpu"))
    a = torch.arange(20, dtype=torch.int32).view(10, 2)
    b = torch.zeros_like(a)
    calls = []

    def proc(i, n, o):
        calls.append(int(i[0].shape[0])); o[0].copy_(i[0] * 2)

    ex.run(proc, np.zeros(10, dtype=np.uint32), [a], [b])
    assert calls == [4, 4, 2] and torch.equal(b, a * 2)
    assert sharding.shard_round_robin(7, 1, 3) == [1, 4] and sharding.chunk_ranges(10, 4) == [(0, 4), (4, 4), (8, 2)]
