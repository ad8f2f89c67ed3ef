#!/usr/bin/env python3
"""bench.py -- LINNE per-frame prediction path on MI355X: frames/s of encode (-m 7) and decode, 44.1 kHz stereo.

One "step" = one pass of the encode hot path (MS, pre-emphasis, LPC analysis + layer cascade for 4 regularisers,
quantisation, int32 FIR cascade: LINNEAmd_EncodeFramesDevice) over one batch of synthetic PCM that is already
resident in HBM.  At N=1 the batch is BASELINE.json configs[1]: 60 min of 44.1 kHz int16 stereo = 15 504 frames of
10 240 samples (the last one a 9 280-sample tail).  With N > 1 every rank encodes its own 60-minute track (frames
shard with no data-path collective): "scaling": "weak".  The decode hot path (configs[2]) is timed right after on
the encode's own output and reported as decode_frames_per_s; it must reproduce the PCM bit for bit.

Starting it.  `python bench.py --gpus N` alone starts the N ranks itself (N child processes, one per GPU, from a parent
that never touches the GPU; rank 0's line is the output; any child's failure is the exit code).  Under a launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) WORLD_SIZE is set and must equal --gpus.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant encode kernel against HBM as BASELINE.json asks (the path is
FP64-VALU / dependent-chain bound; `valu_f64` gives that view), `roofline_decode` the dominant decode kernel;
`cpu_baseline` / `cpu_baseline_decode` time the reference CPU encoder / decoder (oracle/_ref, or the oracle port if absent) on a
bounded sample of the same workload; `block_at_a_time` is the per-call latency of the unchanged 13-symbol API.

Everything behind the timed regions is an optional LEG with a deadline of its own (sample parity, CPU baselines, block-at-a-time
calls, the end-to-end API, the transports).  A leg that raises becomes {"error": ...}; a leg that HANGS is cut off by a watchdog
thread, which prints the line with what there is and ends the process: the headline is never lost to a sick link.  Control
traffic at N > 1 (barriers, the MAX of the times, agreement flags) runs on a gloo group; RCCL carries only what north_star gives
it, the scatter of frame batches and the gather of residuals (transports.rccl_scatter_gather), on a group of its own.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

torch = None            # set by _load(): the launcher parent (--gpus N without WORLD_SIZE) must not import what may touch the GPU
linne_amd = None


def _load():
    global torch, linne_amd
    if torch is None:
        import torch as _torch
        import linne_amd as _linne_amd
        torch, linne_amd = _torch, _linne_amd


ALGO_BYTES_PER_CF = 82552           # 40960 in + 40960 out + 632 params per channel-frame
MAC_PER_CF_REFERENCE = 46.2e6       # as written in the reference (-m 7, N = 10240)
MAC_PER_CF_EXECUTED = 25.2e6        # multiply-adds (separate or fused) per channel-frame in the schedule this build runs; -m 7 only
HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 78.6             # vector FMA peak; unfused mul+add tops out at half of it on paper
FP64_UNFUSED_MEASURED_TFLOPS = 32.4 # what tools/ubench/dp_rate.hip sustains with separate multiply and add (profiles/r01_dp_rate.txt)
ENCODE_KINDS = (1, 3, 4, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16, 18, 19, 20, 21, 22, 23, 25)
DECODE_KINDS = (11, 12, 30, 31, 32, 33, 34, 35, 36)
KERNEL_KINDS = {13: "k_stats", 14: "k_autocorr_lane", 1: "k_prep", 3: "k_autocorr2", 4: "k_levinson_lds",
                5: "k_fir2<2,false,true> (search of the long layer + fused one-unit forward; frames k_search_long does not take)",
                25: "k_search_long<P> (search of the long layer over the shared window + fused one-unit forward)",
                18: "k_fir_small<P,false,*> (search of the last, short layer)", 15: "k_fir_small<P,true,*> (search of layer 0 + fused one-unit forward)",
                6: "k_fir2<0> (exact fallback)", 7: "k_select", 8: "k_fir2<1,false,true> (forward, jobs with several units)",
                19: "k_fir2<1,false,false> (forward of the last layer, frames k_fwd_loss does not take)", 20: "k_last_layer<P> / k_fwd_loss<P> (last layer: exact search + forward pass + ordered loss in one launch / forward pass + loss)", 21: "k_autocorr_hist<P,0> (long layer, one-unit trial)", 22: "k_autocorr_hist<P,1> (long layer, two-unit trial)", 23: "k_autocorr_sub<P> (long layer, trials of order <= 32)", 16: "k_fir2<1,true,*> (forward of layer 0, jobs with several units)",
                9: "k_chain_sum<1>", 10: "k_finalize",      # (k_quantize + k_fir_cascade since round 3; the name keys profiles/pmc_latest.json)
                11: "k_synthesize (one wave per channel-frame, all layers)", 12: "k_ms_to_lr",
                30: "k_synth_big<P> (synthesis of the long layer)", 31: "k_synth_small<P> (synthesis of the short layers, de-emphasis)",
                32: "k_synth_pipe (a wave per stage of the cascade, 16-sample blocks: the latency form)",
                33: "k_synth_rows<NCH> (a long layer: four channel-frames per wave, the old taps on the matrix unit)",
                36: "k_synth_rows8<PB> / k_synth_rows<0> (a short layer: eight / four channel-frames per wave)",
                35: "k_synth_l0_de (layer 0 + de-emphasis + MS -> LR in one launch, tiles in LDS)",
                34: "k_deemph_lr (de-emphasis behind layer 0, MS -> LR on the way out; LINNE_AMD_DECODE_FUSED=0)"}


def synth_track(num_samples, nch, bits, seed, device, rate=44100.0, chunk=1 << 22):
    """compressible synthetic 'music' (SURVEY 8d recipe, generated on the GPU): per channel 6 harmonics of
    110*(ch+1) Hz with amplitudes 0.3/(k+1) and random phases, plus AR(2)-coloured noise; mix, clip, round."""
    _load()
    g = torch.Generator(device=device)
    g.manual_seed(0x4C494E4E ^ seed)
    out = torch.empty((nch, num_samples), dtype=torch.int32, device=device)
    # impulse response of 1 / (1 - 1.6 z^-1 + 0.8 z^-2), 128 taps
    h = [1.0, 1.6]
    for _ in range(126):
        h.append(1.6 * h[-1] - 0.8 * h[-2])
    h = torch.tensor(h[::-1], dtype=torch.float64, device=device).view(1, 1, -1)
    hgain = float(torch.sqrt((h * h).sum()))
    full = float(1 << (bits - 1))
    for ch in range(nch):
        phases = torch.rand(6, generator=g, device=device, dtype=torch.float64) * (2 * np.pi)
        for s0 in range(0, num_samples, chunk):
            n = min(chunk, num_samples - s0)
            t = (torch.arange(s0, s0 + n, device=device, dtype=torch.float64)) / rate
            tone = torch.zeros(n, dtype=torch.float64, device=device)
            for k in range(6):
                tone += (0.3 / (k + 1)) * torch.sin(2 * np.pi * 110.0 * (ch + 1) * (k + 1) * t + phases[k])
            e = torch.randn(n + 127, generator=g, device=device, dtype=torch.float64) * (0.1 / hgain)
            noise = torch.nn.functional.conv1d(e.view(1, 1, -1), h).view(-1)
            x = torch.clamp(0.6 * tone + noise, -0.999, 0.999) * full
            out[ch, s0:s0 + n] = torch.round(x).to(torch.int32)
    return out


def frames_from_track(track, block):
    """[C][num_samples] -> ([F][C][block] zero padded, num_samples per frame)"""
    _load()
    nch, ns = track.shape
    F = (ns + block - 1) // block
    pad = F * block - ns
    if pad:
        track = torch.cat([track, torch.zeros((nch, pad), dtype=track.dtype, device=track.device)], dim=1)
    frames = track.view(nch, F, block).permute(1, 0, 2).contiguous()
    nsm = np.full(F, block, dtype=np.uint32)
    if pad:
        nsm[-1] = block - pad
    return frames, nsm


def host_cores(world=1):
    """CPU threads this process may really use: affinity, capped by the cgroup CPU quota and BENCH_CPU_CORES
    (a one-GPU box grants 16 of the host's cores; with N ranks rank 0 may use the N ranks' share while they wait)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    n = min(n, int(os.environ.get("BENCH_CPU_CORES", str(16 * max(1, world)))))
    return max(1, n)


def _checkers():
    """the CPU checkers under oracle/ (test infrastructure): imported only by the cpu_baseline / sample-parity legs"""
    odir = os.path.join(ROOT, "oracle")
    if odir not in sys.path:
        sys.path.insert(0, odir)
    from bindings import Oracle, REF_SO, EncodeParameter, reference_available
    return Oracle, REF_SO, EncodeParameter, reference_available


def _ptrs(C, x, nch):
    return (C.POINTER(C.c_int32) * nch)(*[x[ch].ctypes.data_as(C.POINTER(C.c_int32)) for ch in range(nch)])


def sample_parity(frames, nsm, res, prm, st, shape, bits, block, preset, ms, nedge=64, nsample_mid=128, threads=None, seed=12345):
    """Outside the timed region: a sample of the timed batch's output -- the first `nedge` frames, the last `nedge` (with the
    ragged tail) and `nsample_mid` random ones in between -- serialised with the host stage (LINNEAmd_PackFrames) and compared
    byte for byte with the CPU reference's EncodeBlock of the same frames (oracle/_ref when built, else the oracle port; a fresh
    handle per frame on both sides).  Equal blocks mean equal pre-emphasis, unit counts, shifts, coefficients and residuals
    (libs/linne_encoder/src/linne_encoder.c:594-752).  decode(encode(x)) == x cannot show that: it holds for any coefficients.
    At N > 1 every rank checks a (smaller) sample of ITS OWN shard and the verdicts are AND-reduced."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from linne_amd.api import LinneApi
    Oracle, REF_SO, EncodeParameter, reference_available = _checkers()
    F, nch, _ = frames.shape
    rng = np.random.default_rng(seed)
    idx = set(range(min(nedge, F))) | set(range(max(0, F - nedge), F))
    if F > 2 * nedge and nsample_mid:
        idx |= set(int(i) for i in rng.choice(np.arange(nedge, F - nedge), size=min(nsample_mid, F - 2 * nedge), replace=False))
    idx = sorted(idx)
    sel = torch.as_tensor(idx, device=frames.device)
    h_pcm, h_res, h_prm, h_st = (t.index_select(0, sel).cpu().numpy() for t in (frames, res, prm, st))
    use_ref = reference_available()
    api = LinneApi(REF_SO) if use_ref else None
    orc = None if use_ref else Oracle()

    def want(k):
        n = int(nsm[idx[k]])
        x = np.ascontiguousarray(h_pcm[k][:, :n])
        if use_ref:
            enc = api.new_encoder(nch, bits, 44100, block, preset, ms)
            out = np.zeros(nch * block * 8 + 65536, dtype=np.uint8)
            osz = C.c_uint32(0)
            r = api.L.LINNEEncoder_EncodeBlock(enc, _ptrs(C, x, nch), n, out.ctypes.data, out.size, C.byref(osz))
            api.L.LINNEEncoder_Destroy(enc)
            assert r == 0
            return out[:osz.value].tobytes()
        enc = orc.encoder(nch, bits, 44100, block, preset, ms)
        b, _, _ = enc.encode_block(x)
        enc.close()
        return b

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads or host_cores()) as ex:
        wanted = list(ex.map(want, range(len(idx))))
    bad = []
    for k, f in enumerate(idx):
        n = np.array([nsm[f]], dtype=np.uint32)
        got, _ = linne_amd.pack_frames(shape, h_pcm[k:k + 1], h_res[k:k + 1], h_prm[k:k + 1], h_st[k:k + 1], n, 0.0, 1)
        if got[0] != wanted[k]:
            bad.append(f)
    return {"ok": not bad, "frames_compared": len(idx), "mismatching_frames": bad[:16], "checker": "reference" if use_ref else "port",
            "what": "LINNEAmd_PackFrames(GPU params/residual/stats) == CPU EncodeBlock bytes, per frame, fresh handle each",
            "seconds": time.perf_counter() - t0}


def _run_threads(work, n):
    ths = [threading.Thread(target=work, args=(t,)) for t in range(n)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return time.perf_counter() - t0


def cpu_baseline(frames_host, bits, rate, block, preset, ms, budget_frames_per_thread, cores=None):
    """reference CPU encoder (EncodeBlock incl. its entropy stage, libs/linne_encoder/src/linne_encoder.c:594-752) on the first
    frames of the workload, one handle per thread over disjoint frames; then the reference CPU DECODER (DecodeBlock,
    libs/linne_decoder/src/linne_decoder.c:564-668) over the blocks the encoder just wrote, again one handle per thread.
    Returns (encode record, decode record or None, the blocks thread 0 wrote in order -- used by the block-at-a-time leg)."""
    import ctypes as C
    from linne_amd.api import LinneApi, _RefDecoderConfig, _RefHeader
    Oracle, REF_SO, EncodeParameter, reference_available = _checkers()
    cores = cores or host_cores()
    F, nch, _ = frames_host.shape
    per = min(budget_frames_per_thread, max(1, F // cores))
    total = per * cores
    if not reference_available():
        o = Oracle()
        p = EncodeParameter(nch, bits, rate, block, preset, int(ms))
        sub = np.ascontiguousarray(frames_host[:total])
        nbytes = C.c_uint64(0)
        dt = o.L.oracle_bench_encode(C.byref(p), sub.ctypes.data, total, cores, C.byref(nbytes))
        return ({"value": total / dt, "unit": "frames/s", "cores": cores, "kind": "port",
                 "sample": f"{total} full {nch}-channel frames of the same track ({per} per thread, {dt:.1f} s wall), EncodeBlock incl. entropy stage",
                 "single_core_frames_per_s": None}, None, None)
    api = LinneApi(REF_SO)
    L = api.L
    blocks = [None] * total

    def enc_work(t):
        enc = api.new_encoder(nch, bits, rate, block, preset, ms)
        out = np.zeros(nch * block * 8 + 65536, dtype=np.uint8)
        osz = C.c_uint32(0)
        for f in range(t * per, (t + 1) * per):
            r = L.LINNEEncoder_EncodeBlock(enc, _ptrs(C, frames_host[f], nch), block, out.ctypes.data, out.size, C.byref(osz))
            assert r == 0
            blocks[f] = out[:osz.value].copy()
        L.LINNEEncoder_Destroy(enc)

    dt = _run_threads(enc_work, cores)
    # one core, a few frames: the per-core rate the speed-up is usually quoted against
    n1 = min(per, 24)
    enc = api.new_encoder(nch, bits, rate, block, preset, ms)
    out = np.zeros(nch * block * 8 + 65536, dtype=np.uint8)
    osz = C.c_uint32(0)
    t1 = time.perf_counter()
    for f in range(n1):
        assert L.LINNEEncoder_EncodeBlock(enc, _ptrs(C, frames_host[f], nch), block, out.ctypes.data, out.size, C.byref(osz)) == 0
    single = n1 / (time.perf_counter() - t1)
    L.LINNEEncoder_Destroy(enc)
    enc_rec = {"value": total / dt, "unit": "frames/s", "cores": cores, "kind": "reference",
               "sample": f"{total} full {nch}-channel frames of the same track ({per} per thread, {dt:.1f} s wall), EncodeBlock incl. entropy stage",
               "single_core_frames_per_s": single}

    # ---- the decoder, on what the encoder wrote
    hdr = _RefHeader(1, 2, nch, total * block, rate, bits, block, preset, int(ms))
    reps = 6
    ok = [True] * cores

    def dec_work(t, first=None, count=None, nrep=reps):
        cfg = _RefDecoderConfig(nch, 5, 128, 1)
        dec = L.LINNEDecoder_Create(C.byref(cfg), None, 0)
        assert dec and L.LINNEDecoder_SetHeader(dec, C.byref(hdr)) == 0
        back = np.zeros((nch, block), dtype=np.int32)
        bp = _ptrs(C, back, nch)
        dsz, dn = C.c_uint32(0), C.c_uint32(0)
        f0 = t * per if first is None else first
        cnt = per if count is None else count
        for rep in range(nrep):
            for f in range(f0, f0 + cnt):
                b = blocks[f]
                if L.LINNEDecoder_DecodeBlock(dec, b.ctypes.data, b.size, bp, nch, block, C.byref(dsz), C.byref(dn)) != 0 or dn.value != block:
                    ok[t] = False
                elif rep == 0 and not np.array_equal(back, frames_host[f]):
                    ok[t] = False
        L.LINNEDecoder_Destroy(dec)

    ddt = _run_threads(dec_work, cores)
    n1d = min(per, 128)
    t1 = time.perf_counter()
    dec_work(0, 0, n1d, 2)
    dsingle = 2 * n1d / (time.perf_counter() - t1)
    dec_rec = {"value": total * reps / ddt, "unit": "frames/s", "cores": cores, "kind": "reference",
               "sample": f"the {total} blocks the reference encoder wrote for cpu_baseline, {reps} passes ({per} blocks per thread, {ddt:.1f} s wall), "
                         "DecodeBlock incl. CRC check and entropy stage", "single_core_frames_per_s": dsingle, "bit_exact": all(ok)}
    return enc_rec, dec_rec, blocks[:per]


def block_at_a_time(frames_host, bits, rate, block, preset, ms, ref_blocks, nblocks=96, warm=8):
    """per-call latency of the reference's own consumers' path: LINNEEncoder_EncodeBlock (tools/linne_codec/linne_codec.c:133-161)
    and LINNEDecoder_DecodeBlock (tools/linne_player/linne_player.c:66-118 calls it from an audio callback) of liblinne_amd.so, one
    block per call, host planes in, bytes out and back -- synchronous, PCIe and the host entropy stage included"""
    import ctypes as C
    from linne_amd.api import LinneApi, _RefDecoderConfig, _RefHeader
    api = LinneApi(linne_amd.LIB_PATH)
    L = api.L
    F, nch, _ = frames_host.shape
    nblocks = min(nblocks, F)
    enc = api.new_encoder(nch, bits, rate, block, preset, ms)
    out = np.zeros(nch * block * 8 + 65536, dtype=np.uint8)
    osz = C.c_uint32(0)
    mine = []
    for f in range(min(warm, nblocks)):     # the first calls create the GPU context and size its buffers
        assert L.LINNEEncoder_EncodeBlock(enc, _ptrs(C, frames_host[f], nch), block, out.ctypes.data, out.size, C.byref(osz)) == 0
    L.LINNEEncoder_Destroy(enc)
    enc = api.new_encoder(nch, bits, rate, block, preset, ms)          # (a fresh handle: the stream's blocks from its first on)
    t0 = time.perf_counter()
    for f in range(nblocks):
        assert L.LINNEEncoder_EncodeBlock(enc, _ptrs(C, frames_host[f], nch), block, out.ctypes.data, out.size, C.byref(osz)) == 0
        mine.append(out[:osz.value].copy())
    enc_ms = (time.perf_counter() - t0) / nblocks * 1e3
    L.LINNEEncoder_Destroy(enc)
    same = None
    if ref_blocks:
        k = min(len(ref_blocks), nblocks)
        same = all(np.array_equal(mine[f], ref_blocks[f]) for f in range(k))
    hdr = _RefHeader(1, 2, nch, nblocks * block, rate, bits, block, preset, int(ms))
    cfg = _RefDecoderConfig(nch, 5, 128, 1)
    dec = L.LINNEDecoder_Create(C.byref(cfg), None, 0)
    assert dec and L.LINNEDecoder_SetHeader(dec, C.byref(hdr)) == 0
    back = np.zeros((nch, block), dtype=np.int32)
    bp = _ptrs(C, back, nch)
    dsz, dn = C.c_uint32(0), C.c_uint32(0)
    ok = True
    for f in range(min(warm, nblocks)):
        assert L.LINNEDecoder_DecodeBlock(dec, mine[f].ctypes.data, mine[f].size, bp, nch, block, C.byref(dsz), C.byref(dn)) == 0
    t0 = time.perf_counter()
    for f in range(nblocks):
        r = L.LINNEDecoder_DecodeBlock(dec, mine[f].ctypes.data, mine[f].size, bp, nch, block, C.byref(dsz), C.byref(dn))
        ok = ok and r == 0 and np.array_equal(back, frames_host[f])
    dec_ms = (time.perf_counter() - t0) / nblocks * 1e3
    L.LINNEDecoder_Destroy(dec)
    return {"encode_ms": enc_ms, "decode_ms": dec_ms, "reference_encode_ms": None, "reference_decode_ms": None, "blocks": nblocks,
            "decode_bit_exact": bool(ok), "bytes_equal_reference_encodeblock": same,
            "what": f"LINNEEncoder_EncodeBlock / LINNEDecoder_DecodeBlock of liblinne_amd.so, one {nch}-channel block of {block} samples per call "
                    "(host planes -> block bytes -> host planes; decode_ms includes the comparison with the input); reference_* = the reference on one host core"}


def end_to_end(x_host, bits, rate, block, preset, ms):
    """the drop-in API on host buffers (host planes -> .lnn bytes -> host planes): LINNEEncoder_EncodeWhole /
    LINNEDecoder_DecodeWhole of liblinne_amd.so, PCIe and the host entropy stage included; second of two runs on the
    same handles (the first pays the one-time set-up of the GPU context and the pinned staging slots)"""
    import ctypes as C
    from linne_amd.api import LinneApi, _planar_ptrs, _RefDecoderConfig
    api = LinneApi(linne_amd.LIB_PATH)
    L = api.L
    nch, ns = x_host.shape
    nf = (ns + block - 1) // block
    enc = api.new_encoder(nch, bits, rate, block, preset, ms)
    cfg = _RefDecoderConfig(nch, 5, 128, 1)
    dec = L.LINNEDecoder_Create(C.byref(cfg), None, 0)
    xp, _k1 = _planar_ptrs(x_host)
    cap = min(x_host.size * 4 + 65536, 0xFFFFFFFF)
    out = np.ones(cap, dtype=np.uint8)
    back = np.ones_like(x_host)
    bp, _k2 = _planar_ptrs(back)
    osz = C.c_uint32(0)
    res = {}
    for rep in range(2):
        t0 = time.perf_counter()
        r1 = L.LINNEEncoder_EncodeWhole(enc, xp, ns, out.ctypes.data, cap, C.byref(osz))
        t1 = time.perf_counter()
        r2 = L.LINNEDecoder_DecodeWhole(dec, out.ctypes.data, osz.value, bp, nch, ns)
        t2 = time.perf_counter()
        res = {"encode_whole_frames_per_s": nf / (t1 - t0), "decode_whole_frames_per_s": nf / (t2 - t1),
               "compression_ratio": osz.value / (x_host.size * (bits // 8)), "round_trip_bit_exact": bool(r1 == 0 and r2 == 0 and np.array_equal(back, x_host)),
               "host_threads": host_cores(), "note": "host planes -> .lnn -> host planes through the 13-symbol API; includes PCIe and the host entropy stage"}
    L.LINNEEncoder_Destroy(enc); L.LINNEDecoder_Destroy(dec)
    return res


def leg_direct_h2d(ctx, shape, frames, nsm, res_ref, steps, chunk_frames, barrier, world, red):
    """Transport "direct": every rank's shard starts in (pinned) HOST memory and its results end there -- H2D of chunk k + 1 and
    D2H of chunk k - 1 on two copy streams beside the kernels of chunk k, each GPU on its own PCIe link, nothing between GPUs
    (the competitor of the RCCL scatter).  Whole-job frames/s incl. both transfers."""
    F, nch, S = frames.shape
    dev = frames.device
    chunks = [(f0, min(chunk_frames, F - f0)) for f0 in range(0, F, chunk_frames)]
    B = chunks[0][1]
    pin = lambda shp, dt: torch.empty(shp, dtype=dt, pin_memory=True)
    h_in = pin((F, nch, S), torch.int32); h_in.copy_(frames)
    h_res = pin((F, nch, S), torch.int32)
    h_prm = pin((F, nch, linne_amd.PARAM_WORDS), torch.int32)
    h_st = pin((F, nch, linne_amd.STAT_WORDS), torch.float64)
    d_in = [torch.empty((B, nch, S), dtype=torch.int32, device=dev) for _ in range(2)]
    d_res = [torch.empty((B, nch, S), dtype=torch.int32, device=dev) for _ in range(2)]
    d_prm = [torch.zeros((B, nch, linne_amd.PARAM_WORDS), dtype=torch.int32, device=dev) for _ in range(2)]
    d_st = [torch.zeros((B, nch, linne_amd.STAT_WORDS), dtype=torch.float64, device=dev) for _ in range(2)]
    s_in, s_out, cur = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream(dev)

    def step():
        in_free, out_done = [None, None], [None, None]
        for k, (f0, cnt) in enumerate(chunks):
            b = k & 1
            with torch.cuda.stream(s_in):
                if in_free[b] is not None:
                    s_in.wait_event(in_free[b])
                d_in[b][:cnt].copy_(h_in[f0:f0 + cnt], non_blocking=True)
                e_in = torch.cuda.Event(); e_in.record(s_in)
            cur.wait_event(e_in)
            if out_done[b] is not None:
                cur.wait_event(out_done[b])
            ctx.encode_frames(shape, d_in[b][:cnt], nsm[f0:f0 + cnt], out=(d_res[b][:cnt], d_prm[b][:cnt], d_st[b][:cnt]))
            e_k = torch.cuda.Event(); e_k.record(cur)
            in_free[b] = e_k
            with torch.cuda.stream(s_out):
                s_out.wait_event(e_k)
                h_res[f0:f0 + cnt].copy_(d_res[b][:cnt], non_blocking=True)
                h_prm[f0:f0 + cnt].copy_(d_prm[b][:cnt], non_blocking=True)
                h_st[f0:f0 + cnt].copy_(d_st[b][:cnt], non_blocking=True)
                e_o = torch.cuda.Event(); e_o.record(s_out)
            out_done[b] = e_o

    step()
    barrier()
    same = bool(torch.equal(h_res, res_ref.cpu()))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = red(time.perf_counter() - t0)
    per_frame = nch * (S * 4 * 2 + linne_amd.PARAM_WORDS * 4 + linne_amd.STAT_WORDS * 8)
    return {"frames_per_s": F * world * steps / dt, "ms_per_step": dt / steps * 1e3, "chunk_frames": B, "chunks_per_step": len(chunks),
            "pcie_gb_per_s_per_gpu_each_way": F * steps * per_frame / 2 / dt / 1e9, "results_equal_resident_path": same,
            "what": "pinned host PCM -> H2D -> encode hot path -> D2H residual+params+stats, per GPU on its own PCIe link, double-buffered"}


def leg_exchange(dist, group, ctx, shape, frames, nsm, res_ref, prm_ref, steps, chunk_frames, barrier, world, rank, red):
    """Transport "scatter / gather": north_star's N > 1 data path.  Rank 0 holds the whole batch (world x this rank's track) in its
    HBM; chunks of frames go round-robin to the ranks by point-to-point send / recv on `group` (RCCL: batched, ncclGroupStart ...
    ncclGroupEnd; a gloo group only in one-GPU rehearsals, where every transfer is staged through host copies), every rank analyses
    its chunks, residual + params + stats come back the same way (linne_amd.sharding.ChunkExchange, software-pipelined: the links
    work while the kernels run; waits are stream waits, the host never blocks on a transfer).  Whole-job frames/s incl. scatter and
    gather.  The record says which backend really moved the bytes."""
    from linne_amd.sharding import ChunkExchange
    F, nch, S = frames.shape
    dev = frames.device
    Ftot = F * world
    nsm_all = np.tile(nsm, world)
    backend = dist.get_backend(group)
    rccl_ranks = None
    if backend == "nccl":           # the whole-group communicator comes up here, inside this leg's deadline
        t = torch.ones(1, device=dev)
        dist.all_reduce(t, group=group)
        torch.cuda.synchronize()
        rccl_ranks = int(t.item())      # what RCCL itself counted: an all-reduce of ones over the communicator
        assert rccl_ranks == world
    ex = ChunkExchange(dist, Ftot, chunk_frames, [((nch, S), torch.int32)],
                       [((nch, S), torch.int32), ((nch, linne_amd.PARAM_WORDS), torch.int32), ((nch, linne_amd.STAT_WORDS), torch.float64)], dev, root=0, group=group)
    if rank == 0:
        all_pcm = frames.repeat(world, 1, 1)
        outs = [torch.empty((Ftot, nch, S), dtype=torch.int32, device=dev), torch.zeros((Ftot, nch, linne_amd.PARAM_WORDS), dtype=torch.int32, device=dev),
                torch.zeros((Ftot, nch, linne_amd.STAT_WORDS), dtype=torch.float64, device=dev)]
        ins = [all_pcm]
    else:
        ins, outs = None, None

    def process(i, n, o):
        ctx.encode_frames(shape, i[0], n, out=tuple(o))

    ex.run(process, nsm_all, ins, outs)
    barrier()
    same = None
    if rank == 0:           # every copy of the track, wherever it was analysed, must equal the resident path's result
        same = all(bool(torch.equal(outs[0][g * F:(g + 1) * F], res_ref)) and bool(torch.equal(outs[1][g * F:(g + 1) * F], prm_ref)) for g in range(world))
    t0 = time.perf_counter()
    for _ in range(steps):
        ex.run(process, nsm_all, ins, outs)
    barrier()
    dt = red(time.perf_counter() - t0)
    moved = (ex.bytes_per_frame_out + ex.bytes_per_frame_back) * Ftot * (world - 1) / max(1, world)
    via = "RCCL ncclSend / ncclRecv over xGMI, device memory to device memory" if backend == "nccl" else f"{backend} with every transfer staged through host copies (a REHEARSAL of the schedule, not an RCCL measurement)"
    return {"frames_per_s": Ftot * steps / dt, "ms_per_step": dt / steps * 1e3, "ranks": world, "rccl_ranks": rccl_ranks, "backend": backend, "staged_through_host": bool(ex.stage_through_host),
            "chunk_frames": chunk_frames, "chunks_per_step": len(ex.chunks), "root_link_gb_per_s": moved * steps / dt / 1e9, "results_equal_resident_path": same,
            "what": f"rank 0's HBM -> scatter of int32 [chunk][C][S] -> encode hot path on every rank -> gather of residual+params+stats to rank 0; pipelined; {via}"}


class Guard:
    """The one JSON line and the deadlines of the optional legs.  run(name, seconds, fn) returns fn()'s record, {"error": ...} if
    it raised; if it has not returned after `seconds`, a watchdog thread records the timeout, prints the line (rank 0) and ends
    the process with a non-zero code (3, or 2 if a parity check had failed before) -- the legs behind a hung one are not run, the
    headline in front of it is never lost, and the run still counts as failed: a hang is a finding, not a result."""

    def __init__(self, rank):
        self.rank, self.line, self.lock, self.printed, self.exit_code = rank, None, threading.Lock(), False, 0

    def emit(self):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if self.rank == 0 and self.line is not None:
                print(json.dumps(self.line), flush=True)

    def run(self, name, seconds, fn, on_timeout=None):
        done = threading.Event()

        def watch():
            if done.wait(seconds):
                return
            rec = {"error": f"timed out after {seconds} s (cut off by the watchdog; the legs behind it were not run)"}
            if self.line is not None:
                if on_timeout is not None:
                    on_timeout(rec)
                else:
                    self.line[name] = rec
                self.line.setdefault("legs_timed_out", []).append(name)
            self.emit()
            sys.stderr.write(f"bench.py rank {self.rank}: leg {name} hung; exiting with a failure code\n")
            sys.stderr.flush()
            os._exit(self.exit_code or 3)       # the line is out, but a leg that hung (on the GPU, on a link) is a FAILURE of the run

        threading.Thread(target=watch, daemon=True).start()
        try:
            out = fn()
        except Exception as exc:
            out = {"error": repr(exc)}
        done.set()
        return out


def launch(n, argv, script=None):
    """`python bench.py --gpus N` with no launcher around it: N child processes, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), started by this parent, which makes no GPU call and never
    replaces itself.  Rank 0's JSON line goes straight to our stdout.  Exit code: the first failing child's; when one fails the
    others get 30 s to finish and are then killed by PID."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env))

    def stop(signum, _frame):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, stop)
    signal.signal(signal.SIGINT, stop)
    rc, deadline = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            c = p.poll()
            if c not in (None, 0) and rc == 0:
                rc, deadline = c, time.time() + 30
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            deadline = None
        time.sleep(0.2)
    for p in procs:
        if p.returncode and rc == 0:
            rc = p.returncode
    return rc


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--minutes", type=float, default=60.0, help="track length per GPU (default: configs[1], 60 min)")
    ap.add_argument("--preset", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames-per-thread", type=int, default=128)
    ap.add_argument("--scratch-gib", type=float, default=30.0)
    ap.add_argument("--channels", type=int, default=2, help="other BASELINE configs, e.g. configs[4]: --channels 8 --bits 24 --rate 96000")
    ap.add_argument("--bits", type=int, default=16)
    ap.add_argument("--rate", type=int, default=44100)
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the EncodeWhole/DecodeWhole leg on host buffers (it launches smaller "
                    "batches: skip it when collecting the per-kernel rocprof summary of the timed region)")
    ap.add_argument("--no-sample-parity", action="store_true", help="skip the sampled comparison of the timed batch's output with the CPU reference")
    ap.add_argument("--no-block-at-a-time", action="store_true", help="skip the EncodeBlock / DecodeBlock per-call latency leg")
    ap.add_argument("--tracks-total", type=int, default=0, help="STRONG scaling: this many tracks in the whole job, split evenly over the GPUs "
                    "(BASELINE configs[3]: --tracks-total 1024 --minutes 3); overrides --tracks")
    ap.add_argument("--no-transports", action="store_true", help="skip the transport legs (direct per-GPU H2D/D2H; RCCL scatter/gather from rank 0)")
    ap.add_argument("--chunk-frames", type=int, default=0, help="frames per pipeline chunk of the transport legs (default: a quarter of the shard)")
    ap.add_argument("--tracks", type=int, default=1, help="tracks per GPU in ONE batch, each --minutes long with its own ragged tail "
                    "(BASELINE configs[3]: --tracks 1024 --minutes 3 on 8 GPUs = 128 per GPU)")
    ap.add_argument("--leg-seconds", type=float, default=60.0, help="deadline of an optional leg (the CPU baselines and the end-to-end leg get twice as long)")
    return ap.parse_args(argv)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch(args.gpus, argv))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={os.environ['WORLD_SIZE']}: start it as `python bench.py --gpus N` "
                 "(it starts the ranks itself) or under torch.distributed.run with --nproc-per-node equal to --gpus")
    worker(args)


def worker(args):
    _load()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    ndev = torch.cuda.device_count()
    local = local % ndev                    # (rehearsals on a one-GPU box put several ranks on the same device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("BENCH_BACKEND", "nccl")       # the transfers' backend: "nccl" is RCCL on ROCm; "gloo" only for one-GPU rehearsals
    guard = Guard(rank)
    dist, xgroup = None, None
    if world > 1:
        import datetime
        import torch.distributed as dist
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # one node: the container's hostname may not resolve
        # control plane (barriers, MAX of the times, agreement flags): gloo over 127.0.0.1 -- nothing the headline needs rides on RCCL
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=240))
        if backend != "nccl":
            xgroup = dist.group.WORLD
        # (backend "nccl": the RCCL group is created INSIDE the exchange leg, under its deadline -- new_group is a collective too)
    scaling = "weak"
    if args.tracks_total:
        assert args.tracks_total % world == 0, "--tracks-total must be a multiple of the GPU count"
        args.tracks = args.tracks_total // world
        scaling = "strong"
    L = args.leg_seconds

    nch, bits, block, rate, ms = args.channels, args.bits, 10240, args.rate, args.channels >= 2
    ns_total = int(round(args.minutes * 60 * rate))
    parts, nparts = [], []
    for t in range(args.tracks):                       # several tracks: one batch, every track with its own ragged tail
        track = synth_track(ns_total, nch, bits, seed=rank * args.tracks + t, device=dev)
        fr, nn = frames_from_track(track, block)
        parts.append(fr); nparts.append(nn)
        del track
    frames = parts[0] if args.tracks == 1 else torch.cat(parts, dim=0)
    nsm = np.concatenate(nparts)
    del parts
    F = frames.shape[0]
    ctx = linne_amd.Context(local, scratch_bytes=int(args.scratch_gib * (1 << 30)))
    shape = ctx.shape(nch, bits, block, args.preset, ms)
    res = torch.empty_like(frames)
    prm = torch.zeros((F, nch, linne_amd.PARAM_WORDS), dtype=torch.int32, device=dev)
    st = torch.zeros((F, nch, linne_amd.STAT_WORDS), dtype=torch.float64, device=dev)
    work = torch.empty_like(frames)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def red_max(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def all_ok(flag):
        if world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def encode_step():
        ctx.encode_frames(shape, frames, nsm, out=(res, prm, st))

    # the decode works in place: every timed step gets a resident copy of the residual of its own, made before the timed region
    # (inputs resident in HBM when it starts, as on the encode side); steps beyond DECODE_COPIES restore theirs inside it
    DECODE_COPIES = 16
    dec_bufs = [work]

    def decode_step(i=0, restore=True):
        buf = dec_bufs[i % len(dec_bufs)]
        if restore:
            buf.copy_(res)
        ctx.decode_frames(shape, buf, prm, nsm)

    for _ in range(args.warmup):
        encode_step()
    barrier()
    ctx.enable_timing(True)
    kern_ms = {k: 0.0 for k in KERNEL_KINDS}
    kern_launches = {k: 0 for k in KERNEL_KINDS}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        encode_step()
        torch.cuda.synchronize()            # per-step sync so the per-kernel events of this step can be read
        for k in ENCODE_KINDS:
            m = ctx.last_ms(k)
            if m > 0:
                kern_ms[k] += m
                kern_launches[k] += ctx.last_launches(k)
    barrier()
    enc_s = red_max(time.perf_counter() - t0)
    ctx.enable_timing(False)

    # One more step, UNTIMED, on ONE compute stream: with the default two streams a kernel's HIP-event span includes what the other
    # half's kernels took of the GPU beside it, so the spans of the timed steps overlap and add up to ~2x the step.  This step's
    # spans are exclusive -- the figures a `LINNE_AMD_STREAMS=1 rocprofv3 --kernel-trace --stats` summary of the same workload
    # gives per kernel (profiles/r0N_kernel_stats_one_stream.csv) -- and they pick and price the roofline's kernel.
    excl_ms = {k: 0.0 for k in KERNEL_KINDS}
    excl_launches = {k: 0 for k in KERNEL_KINDS}
    excl_step_ms = None
    streams_env = os.environ.get("LINNE_AMD_STREAMS")
    os.environ["LINNE_AMD_STREAMS"] = "1"            # (read per call by the library)
    try:
        ctx.enable_timing(True)
        encode_step()
        torch.cuda.synchronize()
        for k in ENCODE_KINDS:
            m = ctx.last_ms(k)
            if m > 0:
                excl_ms[k] = m
                excl_launches[k] = ctx.last_launches(k)
        excl_step_ms = ctx.last_ms(0)            # the whole call, first event to last, on the context's stream
        ctx.enable_timing(False)
    finally:
        if streams_env is None:
            del os.environ["LINNE_AMD_STREAMS"]
        else:
            os.environ["LINNE_AMD_STREAMS"] = streams_env
    barrier()

    # decode: warm-up, then K timed steps
    decode_step()
    barrier()
    ok = bool(torch.equal(work, frames))
    while len(dec_bufs) < min(args.steps, DECODE_COPIES):
        dec_bufs.append(torch.empty_like(work))
    for i in range(1, len(dec_bufs)):            # every copy decoded once before the timed region (a fresh allocation's first pass pays for its pages)
        decode_step(i)
    for b in dec_bufs:
        b.copy_(res)
    barrier()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        decode_step(i, restore=(i >= len(dec_bufs)))
        torch.cuda.synchronize()
        for k in DECODE_KINDS:
            m = ctx.last_ms(k)
            if m > 0:
                kern_ms[k] += m
                kern_launches[k] += ctx.last_launches(k)
    barrier()
    dec_s = red_max(time.perf_counter() - t0)
    ctx.enable_timing(False)
    ok = ok and all(bool(torch.equal(b, frames)) for b in dec_bufs)          # every timed step's output, bit for bit
    del dec_bufs[1:]
    ok = all_ok(ok)
    if not ok:
        guard.exit_code = 2

    # ---- the line: everything the timed regions gave; the legs below fill in the rest
    if rank == 0:
        total_frames = F * world * args.steps
        enc_fps = total_frames / enc_s
        dec_fps = total_frames / dec_s
        layers = linne_amd.PRESET_LAYERS[args.preset]
        nlayers = len(layers)
        n_big = sum(1 for P in layers if P >= 32)
        pmc, pmc_src = {}, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                pmc_src = "profiles/pmc_latest.json (committed file, NOT measured in this run): " + str(pmc.get("_source", ""))
            except Exception:
                pmc = {}

        def roof(kinds, ms, launches_of, nsteps, exclude=()):
            """dominant kernel of `kinds` by summed HIP-event time over `nsteps` steps whose spans do not overlap: algorithmic bytes of
            the channel-frames one launch processes / its average duration, against the HBM peak (DESIGN.md "Measurement")"""
            cand = [k for k in kinds if k not in exclude and ms[k] > 0]
            if not cand:
                return None
            dom = max(cand, key=lambda k: ms[k])
            launches = max(1, launches_of[dom])
            avg_ms = ms[dom] / launches
            # timed spans of one kind per chunk of frames: per-layer kernels have one span per layer they serve
            per_chunk = {25: 1, 3: n_big, 14: nlayers - n_big, 4: nlayers, 5: max(1, nlayers - 2), 15: 1, 6: nlayers, 7: nlayers, 8: max(1, nlayers - 2),
                         16: 1, 21: n_big, 22: n_big, 23: n_big, 30: n_big, 31: nlayers - n_big, 33: n_big, 34: 1, 35: 1, 36: max(1, nlayers - n_big - 1)}.get(dom, 1)
            cf_per_launch = F * nch * nsteps / (launches / per_chunk)
            achieved = ALGO_BYTES_PER_CF * cf_per_launch / (avg_ms * 1e-3) / 1e9
            per_cf = (pmc.get(KERNEL_KINDS[dom]) or {}).get("hbm_bytes_per_channel_frame_per_launch")
            return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": per_cf * cf_per_launch if per_cf else None, "traffic_source": pmc_src if per_cf else None,
                    "kernel": KERNEL_KINDS[dom], "avg_launch_ms": avg_ms, "launches": launches, "channel_frames_per_launch": cf_per_launch}

        # encode: priced on the exclusive spans of the extra one-stream step (see above); k_stats (13) runs beside the analysis
        roofline = roof(ENCODE_KINDS, excl_ms, excl_launches, 1, exclude=(13,))
        if roofline:
            roofline["measured_on"] = ("one extra untimed encode step of the same batch with LINNE_AMD_STREAMS=1 (exclusive per-kernel HIP-event spans on the "
                                       "launch stream); agrees with the rocprofv3 --kernel-trace --stats summary of the same workload on one stream "
                                       "(profiles/r04_kernel_stats_one_stream.csv), and with twice the half-batch launches of the two-stream summary")
            roofline["exclusive_kernel_ms"] = {KERNEL_KINDS[k]: round(excl_ms[k], 3) for k in ENCODE_KINDS if excl_ms[k] > 0}
            roofline["one_stream_step_ms"] = round(excl_step_ms, 3) if excl_step_ms and excl_step_ms > 0 else None
            roofline["exclusive_kernel_ms_note"] = ("k_stats and, for a ragged tail, k_autocorr2 run on the side stream BESIDE the kernels of the main one: "
                                                    "their spans are not part of the sum that makes one_stream_step_ms")
            roofline["note"] = ("algorithmic bytes = 82552 B per channel-frame; the kernel is FP64-VALU/latency bound, see valu_f64 "
                                "(which rests on a CONSTANT multiply-add count per channel-frame, not on a counter)")
        whole = (pmc.get("_whole_step") or {})
        if roofline and whole.get("encode_hbm_bytes_per_channel_frame"):
            roofline["whole_step_traffic_per_channel_frame"] = whole["encode_hbm_bytes_per_channel_frame"]
            roofline["whole_step_traffic_over_algorithmic"] = whole["encode_hbm_bytes_per_channel_frame"] / ALGO_BYTES_PER_CF
        roofline_decode = roof(DECODE_KINDS, kern_ms, kern_launches, args.steps)
        if roofline_decode:
            roofline_decode["note"] = ("algorithmic bytes = 82552 B per channel-frame (residual in, PCM out, parameters) per launch: a launch of the cascade "
                                       "streams the channel-frame in and out once; k_synth_rows is bound by the vector unit's issue rate (a dependent "
                                       "recurrence per channel-frame, four channel-frames per wave), not by HBM; frac is the DOMINANT launch's, "
                                       "per_kernel lists every launch, whole_step_frac prices the step")
            if whole.get("decode_hbm_bytes_per_channel_frame"):
                roofline_decode["whole_step_traffic_per_channel_frame"] = whole["decode_hbm_bytes_per_channel_frame"]
            # the whole decode step against the same peak: the cascade's layers are launches of their own (each streams the
            # channel-frame in and out once), so the step moves the algorithmic bytes once in nlayers + 1 passes
            roofline_decode["whole_step_achieved"] = ALGO_BYTES_PER_CF * F * nch / (dec_s / args.steps) / 1e9
            roofline_decode["whole_step_frac"] = roofline_decode["whole_step_achieved"] / HBM_PEAK_GBS
            # every launch of the decode step by itself (the dominant one is `kernel` above): each streams the channel-frame in and out once
            per = {}
            for k in DECODE_KINDS:
                if kern_launches[k] > 0 and kern_ms[k] > 0:
                    avg = kern_ms[k] / kern_launches[k]
                    per[KERNEL_KINDS[k]] = {"avg_launch_ms": round(avg, 4), "launches_per_step": kern_launches[k] / args.steps,
                                            "frac": round(ALGO_BYTES_PER_CF * F * nch / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            roofline_decode["per_kernel"] = per
        cf_per_s = enc_fps * nch / world        # per GPU
        valu = {"executed_tflops": 2 * MAC_PER_CF_EXECUTED * cf_per_s / 1e12, "reference_equiv_tflops": 2 * MAC_PER_CF_REFERENCE * cf_per_s / 1e12,
                "peak_fma_tflops": FP64_PEAK_TFLOPS, "frac_of_unfused_peak": 2 * MAC_PER_CF_EXECUTED * cf_per_s / 1e12 / (FP64_PEAK_TFLOPS / 2),
                "measured_unfused_mul_add_tflops": FP64_UNFUSED_MEASURED_TFLOPS,
                "frac_of_measured_unfused": 2 * MAC_PER_CF_EXECUTED * cf_per_s / 1e12 / FP64_UNFUSED_MEASURED_TFLOPS,
                "note": "values that reach the stream may not fuse multiply and add (the certified search may); tools/ubench/dp_rate.hip "
                        "sustains 32.4 TFLOP/s of unfused FP64 mul+add on this GPU (profiles/r01_dp_rate.txt), 63 with FMA; the multiply-add "
                        "count per channel-frame is DESIGN.md's figure for -m 7 (a constant, not a counter)"} if args.preset == 7 else None
        breakdown = {KERNEL_KINDS[k]: round(kern_ms[k] / args.steps, 3) for k in KERNEL_KINDS if kern_ms[k] > 0}
        guard.line = {
            "metric": "frames/sec encode (-m 7) + decode, 44.1 kHz stereo, bit-exact; 1/2/4/8 GPU",
            "value": enc_fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": enc_s / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"encode -m {args.preset}: {args.tracks} x {args.minutes:g} min {rate / 1000:g} kHz int{bits} {nch}-channel per GPU = {F} frames of {block} samples "
                                   f"(tail {int(nsm[-1])}), MS {'on' if ms else 'off'}, PCM resident in HBM; decode = inverse hot path on the encode output",
                       "frames_per_gpu": F, "channels": nch, "preset": args.preset, "tracks_per_gpu": args.tracks, "sharding": f"{world * args.tracks} independent track(s), {args.tracks} per GPU"},
            "ranks": world, "control_backend": "gloo" if world > 1 else None, "transfer_backend": (backend if world > 1 else None),
            "decode_frames_per_s": dec_fps, "decode_ms_per_step": dec_s / args.steps * 1e3, "decode_bit_exact": ok,
            "decode_input": f"the decode is in place: each of the first {DECODE_COPIES} timed steps decodes a resident copy of the residual made before the timed region (later steps restore theirs inside it); every step's output is compared with the PCM",
            "encode_sample_parity": None, "encode_sample_parity_detail": None,
            "encode_channel_frames_per_s": enc_fps * nch,
            "kernel_ms_per_step": breakdown,
            "kernel_ms_note": ("summed HIP-event time of each kernel kind per step.  A batch of this size runs as two halves on two compute streams "
                               "(LINNE_AMD_STREAMS, default 2 when each half keeps the large-batch kernel forms): a kernel's time includes what the other "
                               "half's kernels took of the GPU beside it, so the encode kinds add up to about twice ms_per_step"
                               if os.environ.get("LINNE_AMD_STREAMS", "2") != "1" else "summed HIP-event time of each kernel kind per step (one compute stream)"),
            "roofline": roofline, "roofline_decode": roofline_decode, "valu_f64": valu, "cpu_baseline": None, "cpu_baseline_decode": None,
            "block_at_a_time": None, "end_to_end_api": None, "transports": None, "rccl_ranks": None, "n1_reference": None, "value_over_n1": None,
            "transports_note": "value = the hot path with every rank's shard resident in its own HBM (contract: inputs resident when the timed "
                               "region starts); transports = the same work with the data starting elsewhere: in pinned host memory of each rank "
                               "(direct_h2d) or in rank 0's HBM (rccl_scatter_gather, N > 1), transfers inside the timed region",
        }
    line = guard.line

    # ---- leg: sampled parity against the CPU reference, on EVERY rank's own shard
    if not args.no_sample_parity:
        small = world > 1
        par = guard.run("encode_sample_parity_detail", L, lambda: sample_parity(
            frames, nsm, res, prm, st, shape, bits, block, args.preset, ms, nedge=8 if small else 64, nsample_mid=16 if small else 128,
            threads=max(1, host_cores() // world) if small else None, seed=12345 + rank))
        mine_ok = bool(par.get("ok"))
        everyone = guard.run("encode_sample_parity_agreement", L, lambda: {"ok": all_ok(mine_ok)})
        if rank == 0:
            line["encode_sample_parity"] = bool(everyone.get("ok"))
            if world > 1:
                par["ranks_checked"] = world
                par["what"] = par.get("what", "") + f"; every rank checked {par.get('frames_compared')} frames of its own shard, verdicts AND-reduced"
            line["encode_sample_parity_detail"] = par
        if not everyone.get("ok"):
            guard.exit_code = 2

    # ---- legs on rank 0: the CPU baselines at EVERY N (north_star: the reference's CPU path timed on the box's own host cores in the
    # same run as the 1 / 2 / 4 / 8 GPU figures; the other ranks wait at the barrier below, their GPUs idle -- nothing is timed then),
    # the block-at-a-time API and the end-to-end API at N = 1
    if rank == 0:
        ref_blocks = None
        host_frames = None
        cores = host_cores(world)
        if not args.no_cpu_baseline or (world == 1 and not args.no_block_at_a_time):
            nf = min(F - 1, max(64, cores * args.cpu_frames_per_thread))
            host_frames = frames[:nf].cpu().numpy()
        if not args.no_cpu_baseline:
            out = guard.run("cpu_baseline", 2 * L, lambda: cpu_baseline(host_frames, bits, rate, block, args.preset, ms, args.cpu_frames_per_thread, cores))
            if isinstance(out, dict):
                line["cpu_baseline"] = out
            else:
                line["cpu_baseline"], line["cpu_baseline_decode"], ref_blocks = out
    if world > 1 and not args.no_cpu_baseline:
        guard.run("cpu_baseline_wait", 2 * L + 30, lambda: dist.barrier() or {})
    if rank == 0 and world == 1:
        if not args.no_block_at_a_time:
            bat = guard.run("block_at_a_time", L, lambda: block_at_a_time(host_frames, bits, rate, block, args.preset, ms, ref_blocks))
            cpu, cpud = line.get("cpu_baseline") or {}, line.get("cpu_baseline_decode") or {}
            if "error" not in bat:
                if cpu.get("single_core_frames_per_s"):
                    bat["reference_encode_ms"] = 1e3 / cpu["single_core_frames_per_s"]
                if cpud.get("single_core_frames_per_s"):
                    bat["reference_decode_ms"] = 1e3 / cpud["single_core_frames_per_s"]
            line["block_at_a_time"] = bat
        del host_frames
        if not args.no_end_to_end and args.tracks == 1:
            def e2e():
                x_host = np.ascontiguousarray(frames.permute(1, 0, 2).reshape(nch, -1)[:, :ns_total].cpu().numpy())
                return end_to_end(x_host, bits, rate, block, args.preset, ms)
            line["end_to_end_api"] = guard.run("end_to_end_api", 2 * L, e2e)

    # ---- legs: the transports
    if not args.no_transports:
        transports = {}
        if rank == 0:
            line["transports"] = transports
        chunk = args.chunk_frames or (F + 3) // 4
        tsteps = max(1, min(args.steps, 3))

        def put(name):
            return lambda rec: transports.__setitem__(name, rec)

        n1 = None
        if world > 1:
            # the N = 1 figures of the SAME legs, measured by rank 0 alone while the other ranks wait (their GPUs idle): what
            # `over_n1` below divides by, so the line itself says how the paths that move data scale from 1 to N GPUs
            def alone():
                sync = torch.cuda.synchronize
                for _ in range(max(1, args.warmup)):
                    encode_step()
                sync()
                t1 = time.perf_counter()
                for _ in range(tsteps):
                    encode_step()
                sync()
                resident = F * tsteps / (time.perf_counter() - t1)
                d = leg_direct_h2d(ctx, shape, frames, nsm, res, tsteps, chunk, sync, 1, lambda v: v)
                return {"resident_frames_per_s": resident, "direct_h2d_frames_per_s": d.get("frames_per_s"), "steps": tsteps,
                        "what": "rank 0 alone (the other ranks wait, their GPUs idle): the resident hot path and the direct_h2d leg at N = 1, same shard, same chunking"}
            if rank == 0:
                n1 = guard.run("n1_reference", L, alone)
                line["n1_reference"] = n1
                if n1.get("resident_frames_per_s"):
                    line["value_over_n1"] = line["value"] / n1["resident_frames_per_s"]
            guard.run("n1_reference_wait", L + 30, lambda: dist.barrier() or {})

        transports["direct_h2d"] = guard.run("transports.direct_h2d", L, lambda: leg_direct_h2d(ctx, shape, frames, nsm, res, tsteps, chunk, barrier, world, red_max),
                                             on_timeout=put("direct_h2d"))
        if world > 1:
            # the exchange comes LAST: should RCCL hang, everything else is already in the line.  Every rank agrees on the outcome
            # over the gloo group before anyone believes the record; after a failure nothing touches the RCCL group again.
            name = "rccl_scatter_gather" if backend == "nccl" else f"{backend}_scatter_gather_rehearsal"
            def exchange():
                import datetime
                g = xgroup if xgroup is not None else dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
                return leg_exchange(dist, g, ctx, shape, frames, nsm, res, prm, tsteps, chunk, barrier, world, rank, red_max)
            rec = guard.run("transports." + name, 1.5 * L, exchange, on_timeout=put(name))
            agreed = guard.run("transports." + name + ".agreement", L, lambda: {"ok": all_ok("error" not in rec)}, on_timeout=put(name))
            if not agreed.get("ok") and "error" not in rec:
                rec = {"error": "another rank failed or timed out in this leg", "this_rank": rec}
            transports[name] = rec
            if rank == 0:
                line["rccl_ranks"] = rec.get("rccl_ranks") if isinstance(rec, dict) else None
        if rank == 0 and n1 and "error" not in n1:
            for name, rec in transports.items():
                if isinstance(rec, dict) and rec.get("frames_per_s"):
                    base = n1.get("direct_h2d_frames_per_s") if name == "direct_h2d" else n1.get("resident_frames_per_s")
                    if base:
                        rec["over_n1"] = rec["frames_per_s"] / base
                        rec["over_n1_base"] = ("n1_reference.direct_h2d_frames_per_s" if name == "direct_h2d" else
                                               "n1_reference.resident_frames_per_s (with one rank the scatter / gather has nobody to talk to: the leg IS the resident path)")
        torch.cuda.empty_cache()

    if rank == 0:
        cpu, e2e = line.get("cpu_baseline"), line.get("end_to_end_api")
        if cpu and cpu.get("value"):
            # like for like: the reference's EncodeBlock starts from host memory and includes its entropy coder, so the honest
            # ratio is the drop-in API's EncodeWhole (host planes -> .lnn bytes); the HBM-resident hot path alone is given beside it
            if e2e and e2e.get("encode_whole_frames_per_s"):
                line["speedup_end_to_end_vs_cpu_baseline"] = e2e["encode_whole_frames_per_s"] / cpu["value"]
            line["hot_path_resident_over_cpu_encodeblock"] = line["value"] / cpu["value"]
        cpud = line.get("cpu_baseline_decode")
        if cpud and cpud.get("value"):
            if e2e and e2e.get("decode_whole_frames_per_s"):
                line["speedup_decode_end_to_end_vs_cpu_baseline"] = e2e["decode_whole_frames_per_s"] / cpud["value"]
            line["decode_hot_path_resident_over_cpu_decodeblock"] = line["decode_frames_per_s"] / cpud["value"]
    guard.emit()
    ctx.close()
    if world > 1:
        # leave together; the process groups are not torn down collectively (a communicator that has seen a failed leg may hang in
        # its destructor) -- the line is out, the processes just end
        guard.run("final_barrier", 30, lambda: dist.barrier() or {})
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(guard.exit_code)
    if guard.exit_code:
        sys.exit(guard.exit_code)


if __name__ == "__main__":
    main()
