"""GPU parity at the batch sizes the launch rules pick BY THEMSELVES (-m gpu).

EncodeFramesDevice chooses its kernel forms by batch size (lnn_device.hip: k_autocorr_hist / k_autocorr_sub from 12 288
jobs, k_fwd_loss from 24 576 jobs; DecodeFramesDevice: k_synth_rows + k_deemph_lr from 1 536 channel-frames, k_synth_rows8 for the
short layers from 20 480), cuts a call
into equal chunks when the scratch arena is small, and can rotate chunks over two streams.  The tests in
test_gpu_parity.py reach those forms by forcing them onto a few frames; here the batch is big enough that nothing is
forced: many full 64-row blocks per class run, a ragged tail, several chunks, two streams, and the many-tracks shape of
BASELINE configs[3] (full ... tail, full ... tail in ONE call).  A sample of the output -- first 64 frames, last 64 incl.
the tail, random frames in between -- is compared with the oracle bit for bit (params, stats, residual:
libs/linne_encoder/src/linne_encoder.c:594-752), the whole batch through decode(encode(x)) == x.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import linne_amd
from signals import music
from test_gpu_parity import _check_taps

pytestmark = pytest.mark.gpu

NCH, BITS, BLOCK, PRESET = 2, 16, 10240, 7
NBIG = 3200                     # J = 3200 * 2 * 4 = 25 600 jobs: above both thresholds; 6 400 channel-frames: above decode's
TAIL = 9280


@pytest.fixture(scope="module")
def big(oracle):
    """3200 stereo frames of one long 'music' track, the last one a 9280-sample tail, + a cache of oracle results"""
    x = music(NCH, (NBIG - 1) * BLOCK + TAIL, BITS, seed=2026)
    frames = np.zeros((NBIG, NCH, BLOCK), dtype=np.int32)
    flat = np.zeros((NCH, NBIG * BLOCK), dtype=np.int32)
    flat[:, :x.shape[1]] = x
    frames[:] = flat.reshape(NCH, NBIG, BLOCK).transpose(1, 0, 2)
    ns = np.full(NBIG, BLOCK, dtype=np.uint32)
    ns[-1] = TAIL
    return {"frames": frames, "ns": ns, "cache": {}}


def sample_indices(F, seed, nmid=128):
    rng = np.random.default_rng(seed)
    idx = set(range(min(64, F))) | set(range(max(0, F - 64), F))
    if F > 128:
        idx |= set(int(i) for i in rng.choice(np.arange(64, F - 64), size=min(nmid, F - 128), replace=False))
    return sorted(idx)


def oracle_taps(oracle, frames, ns, indices, cache=None, preset=PRESET, nch=NCH, bits=BITS, block=BLOCK, ms=True):
    """oracle hot path of the sampled frames on the host's cores (ctypes releases the GIL)"""
    cache = {} if cache is None else cache
    todo = [f for f in indices if (f, int(ns[f])) not in cache]

    def one(f):
        enc = oracle.encoder(nch, bits, 44100, block, preset, ms)
        tap, res = enc.hotpath(frames[f][:, :int(ns[f])])
        enc.close()
        return f, tap, res

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        for f, tap, res in ex.map(one, todo):
            cache[(f, int(ns[f]))] = (tap, res)
    return {f: cache[(f, int(ns[f]))] for f in indices}


def compare_sample(want, ns, res, prm, st, where, preset=PRESET, nch=NCH):
    for f, (tap, ores) in want.items():
        n = int(ns[f])
        _check_taps(tap, prm[f], st[f], preset, nch, f"{where} frame {f} n={n}")
        assert np.array_equal(ores, res[f][:, :n]), f"{where} frame {f}: residual"
        assert not res[f][:, n:].any(), f"{where} frame {f}: residual beyond the frame's length"


def run_batch(c, frames, ns, check_decode=True, preset=PRESET, nch=NCH, bits=BITS, block=BLOCK):
    import torch
    shape = c.shape(nch, bits, block, preset, True)
    pcm = torch.from_numpy(frames).cuda()
    res, prm, st = c.encode_frames(shape, pcm, ns)
    c.synchronize()
    c.launches = {k: c.last_launches(k) for k in (1, 3, 18, 20, 21, 22, 23)}        # (the decode call below resets the spans)
    out = (res.cpu().numpy(), prm.cpu().numpy(), st.cpu().numpy())
    if check_decode:
        dec = c.decode_frames(shape, res, prm, ns)
        c.synchronize()
        back = dec.cpu().numpy()
        for f in np.flatnonzero(ns < block):
            back[f, :, int(ns[f]):] = frames[f, :, int(ns[f]):]
        assert np.array_equal(back, frames), "decode(encode(x)) != x"
    del pcm, res, prm, st
    torch.cuda.empty_cache()
    return out


def kinds_that_ran(c, kinds):
    return {k for k in kinds if c.launches.get(k, 0) > 0}


def test_large_batch_picks_the_lanes_kernels_by_itself(oracle, big):
    """3200 stereo frames + tail in ONE chunk: k_autocorr_hist<128,0/1>, k_autocorr_sub, autocorr_rows, k_fwd_loss,
    k_synth_small/big are what the batch-size rules select; nothing is forced"""
    for v in ("LINNE_AMD_HIST", "LINNE_AMD_FWD_LOSS", "LINNE_AMD_ROWS16", "LINNE_AMD_L0_PRODUCTS", "LINNE_AMD_DECODE_KERNEL", "LINNE_AMD_DECODE_ROWS8", "LINNE_AMD_SORT"):
        assert v not in os.environ
    c = linne_amd.Context(0, scratch_bytes=8 << 30, use_torch_stream=False)
    try:
        c.enable_timing(True)
        res, prm, st = run_batch(c, big["frames"], big["ns"], check_decode=False)
        ran = kinds_that_ran(c, (20, 21, 22, 23, 3))
        assert {20, 21, 22, 23} <= ran, f"the lanes = jobs kernels did not run: {ran}"
        assert c.launches[21] == 1, "expected one chunk"
        assert c.last_fallback_count() == 0
        margin = c.last_min_margin()
        assert 0.0 < margin < 1e300
        print(f"min certified margin over {NBIG * NCH * 4 * 3} searches: {margin:.3e}")
        c.enable_timing(False)
        idx = sample_indices(NBIG, seed=1)
        compare_sample(oracle_taps(oracle, big["frames"], big["ns"], idx, big["cache"]), big["ns"], res, prm, st, "one chunk")
        big["one_chunk"] = (res, prm, st)
        run_batch(c, big["frames"], big["ns"], check_decode=True)           # the decode kernels of the large batch
    finally:
        c.close()


def test_multi_chunk_loop(oracle, big):
    """the same call through an arena that holds a fifth of it: chunks with f0 > 0 (small forms by batch size), every frame
    equal to the one-chunk result, sample equal to the oracle"""
    c = linne_amd.Context(0, scratch_bytes=int(1.3 * (1 << 30)), use_torch_stream=False)
    try:
        c.enable_timing(True)
        res, prm, st = run_batch(c, big["frames"], big["ns"], check_decode=False)
        assert c.launches[1] >= 3, f"expected >= 3 chunks, got {c.launches[1]}"
        c.enable_timing(False)
    finally:
        c.close()
    idx = sample_indices(NBIG, seed=1)
    compare_sample(oracle_taps(oracle, big["frames"], big["ns"], idx, big["cache"]), big["ns"], res, prm, st, "multi-chunk")
    if "one_chunk" in big:
        a = big["one_chunk"]
        assert np.array_equal(a[0], res) and np.array_equal(a[1], prm) and np.array_equal(a[2], st, equal_nan=True)


def test_multi_chunk_loop_with_the_large_batch_forms(oracle, big, monkeypatch):
    """several chunks AND the lanes = jobs kernels in every one of them (forced: a chunk of this size would not pick them)"""
    monkeypatch.setenv("LINNE_AMD_HIST", "1")
    monkeypatch.setenv("LINNE_AMD_FWD_LOSS", "1")
    F = 900
    frames, ns = big["frames"][-F:], big["ns"][-F:]
    c = linne_amd.Context(0, scratch_bytes=int(0.6 * (1 << 30)), use_torch_stream=False)
    try:
        c.enable_timing(True)
        res, prm, st = run_batch(c, frames, ns, check_decode=False)
        assert c.launches[1] >= 3 and c.launches[21] == c.launches[1] and c.launches[20] == c.launches[1], c.launches
        c.enable_timing(False)
    finally:
        c.close()
    off = NBIG - F
    idx = sample_indices(F, seed=3, nmid=64)
    want = oracle_taps(oracle, big["frames"], big["ns"], [off + f for f in idx], big["cache"])
    compare_sample({f - off: v for f, v in want.items()}, ns, res, prm, st, "multi-chunk, forced forms")


def test_two_streams(oracle, big, monkeypatch):
    """LINNE_AMD_STREAMS=2: chunks rotate over two streams with their own halves of the arena"""
    monkeypatch.setenv("LINNE_AMD_STREAMS", "2")
    c = linne_amd.Context(0, scratch_bytes=8 << 30, use_torch_stream=False)
    try:
        c.enable_timing(True)
        res, prm, st = run_batch(c, big["frames"], big["ns"], check_decode=False)
        assert c.launches[1] >= 2, "expected the call to split over the two streams"
        c.enable_timing(False)
    finally:
        c.close()
    idx = sample_indices(NBIG, seed=1)
    compare_sample(oracle_taps(oracle, big["frames"], big["ns"], idx, big["cache"]), big["ns"], res, prm, st, "two streams")
    if "one_chunk" in big:
        a = big["one_chunk"]
        assert np.array_equal(a[0], res) and np.array_equal(a[1], prm) and np.array_equal(a[2], st, equal_nan=True)


def many_tracks(big, ntracks, per_track, tail):
    """BASELINE configs[3] in small: ntracks tracks of (per_track - 1) full frames + one tail frame, back to back"""
    F = ntracks * per_track
    frames = big["frames"][:F].copy()
    ns = np.full(F, BLOCK, dtype=np.uint32)
    for t in range(ntracks):
        f = t * per_track + per_track - 1
        ns[f] = tail
        frames[f, :, tail:] = 0
    return frames, ns


@pytest.mark.parametrize("two", ["0", "1"])
def test_long_layer_search_in_one_pass_and_in_one_pass_per_trial(oracle, big, monkeypatch, two):
    """k_search_long's two forms (LINNE_AMD_SEARCH_TWO): the joining trials on the one-unit trial's window registers (four waves
    per SIMD), or a pass per big trial (five) -- the same trial sums up to the certificate's slack, the same forward output bit
    for bit: sampled frames of the 3200-frame batch against the oracle, and the two forms against each other"""
    monkeypatch.setenv("LINNE_AMD_SEARCH_TWO", two)
    c = linne_amd.Context(0, scratch_bytes=8 << 30, use_torch_stream=False)
    try:
        res, prm, st = run_batch(c, big["frames"], big["ns"], check_decode=False)
    finally:
        c.close()
    idx = sample_indices(NBIG, seed=3)
    compare_sample(oracle_taps(oracle, big["frames"], big["ns"], idx, big["cache"]), big["ns"], res, prm, st, f"search form {two}")
    if "one_chunk" in big:
        a = big["one_chunk"]
        assert np.array_equal(a[0], res) and np.array_equal(a[1], prm) and np.array_equal(a[2], st, equal_nan=True)


def test_many_tracks_in_one_call_keep_the_fast_kernels(oracle, big):
    """64 tracks x (49 full + tail 2000) in ONE EncodeFramesDevice call (tools/linne_codec/linne_codec.c:133-161 per track):
    the host sorts the frames by length class, so the batch has two class runs and the lanes = jobs kernels serve the full
    frames exactly as in the single-track run; bytes equal the oracle's on a sample that holds every tail"""
    ntracks, per = 64, 50
    frames, ns = many_tracks(big, ntracks, per, 2000)
    F = ntracks * per
    c = linne_amd.Context(0, scratch_bytes=8 << 30, use_torch_stream=False)
    try:
        c.enable_timing(True)
        res, prm, st = run_batch(c, frames, ns, check_decode=True)
        ran = kinds_that_ran(c, (20, 21, 22, 23))
        assert ran == {20, 21, 22, 23}, f"many-track batch fell off the fast path: {c.launches}"
        assert c.launches[21] == 1
        c.enable_timing(False)
    finally:
        c.close()
    tails = [t * per + per - 1 for t in range(ntracks)]
    rng = np.random.default_rng(5)
    idx = sorted(set(tails) | set(int(i) for i in rng.choice(F, size=96, replace=False)) | set(range(8)))
    cache = {}
    compare_sample(oracle_taps(oracle, frames, ns, idx, cache), ns, res, prm, st, "many tracks")


@pytest.mark.parametrize("forced", [False, True])
def test_unsorted_alternating_lengths_take_the_mixed_run_fallback(ctx_env, oracle, forced):
    """LINNE_AMD_SORT=0 (test knob): frame lengths alternate, more runs than the run table holds, blocks mix classes.  The
    general kernels must serve every row -- also with k_fwd_loss / autocorr_rows forced on (LINNE_AMD_FWD_LOSS=1,
    LINNE_AMD_ROWS16=1, LINNE_AMD_L0_PRODUCTS=0), which a batch of this size would not pick"""
    env = {"LINNE_AMD_SORT": "0"}
    if forced:
        env.update({"LINNE_AMD_FWD_LOSS": "1", "LINNE_AMD_ROWS16": "1", "LINNE_AMD_L0_PRODUCTS": "0", "LINNE_AMD_HIST": "1"})
    lens = [10240, 8192, 6144] * 6                              # 18 runs > LNN_MAXRUN = 16
    frames = np.zeros((len(lens), NCH, BLOCK), dtype=np.int32)
    for f, n in enumerate(lens):
        frames[f, :, :n] = music(NCH, n, BITS, seed=700 + f)
    ns = np.array(lens, dtype=np.uint32)
    with ctx_env(env) as c:
        res, prm, st = run_batch(c, frames, ns, check_decode=True)
    compare_sample(oracle_taps(oracle, frames, ns, list(range(len(lens)))), ns, res, prm, st, f"unsorted, forced={forced}")
    # and the same batch in sorted order (default): same bytes
    with ctx_env({k: v for k, v in env.items() if k != "LINNE_AMD_SORT"}) as c:
        r2 = run_batch(c, frames, ns, check_decode=False)
    assert np.array_equal(r2[0], res) and np.array_equal(r2[1], prm) and np.array_equal(r2[2], st, equal_nan=True)


def test_exact_search_everywhere_equals_the_certified_search(ctx_env, big):
    """LINNE_AMD_EXACT=1 sends every unit-count search through the ordered unfused chains (k_fir2<0>); the certified
    order-free search must give the same bytes on every frame (linne_network.c:338-341)"""
    F = 192
    frames, ns = big["frames"][-F:], big["ns"][-F:]
    with ctx_env({}) as c:
        a = run_batch(c, frames, ns, check_decode=False)
        assert c.last_fallback_count() == 0
        margin = c.last_min_margin()
    with ctx_env({"LINNE_AMD_EXACT": "1"}) as c:
        b = run_batch(c, frames, ns, check_decode=False)
        assert c.last_fallback_count() == F * NCH * 4 * 3
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
    assert margin > 0.0


def test_last_layer_in_one_launch_equals_the_three_kernel_form(ctx_env, oracle, big):
    """A chunk of 49 152 jobs or more (LINNE_AMD_LAST_LAYER=2: 24 576) whose frames all have every trial: k_last_layer makes the last layer's exact ordered search means,
    the strict-< argmin and the winner's forward loss in ONE pass over the input (lanes = jobs), where LINNE_AMD_LAST_LAYER=0 runs the
    certified search (k_fir_small), the selection with its fallback and k_fwd_loss.  Same parameters, residual and statistics (the loss
    of the best pass among them) on every frame of the batch, tail included; a sample of them against the oracle"""
    with ctx_env({"LINNE_AMD_LAST_LAYER": "2"}, scratch_bytes=8 << 30) as c:        # (2: from 24 576 jobs on; the default waits for 49 152)
        c.enable_timing(True)
        a = run_batch(c, big["frames"], big["ns"], check_decode=False)
        assert c.launches[20] >= 1 and c.launches[18] == 0, f"k_last_layer did not take the chunk: {c.launches}"
        assert c.last_fallback_count() == 0
    with ctx_env({"LINNE_AMD_LAST_LAYER": "0"}, scratch_bytes=8 << 30) as c:
        c.enable_timing(True)
        b = run_batch(c, big["frames"], big["ns"], check_decode=False)
        assert c.launches[18] >= 1, c.launches
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
    idx = sample_indices(NBIG, seed=7)
    compare_sample(oracle_taps(oracle, big["frames"], big["ns"], idx, big["cache"]), big["ns"], a[0], a[1], a[2], "k_last_layer")


def test_fwd_loss_forms_agree(ctx_env, big):
    """the last layer's forward pass and loss for a chunk of 25 600 jobs: five waves per 64 jobs (k_fwd_loss_mw: four filter 16 samples
    of a super-tile each, the fifth adds the magnitudes in sample order -- the default below 65 536 jobs) and one wave per 64 jobs
    (LINNE_AMD_FWD_LOSS_MW=0) give the same statistics -- the best pass and its loss among them --, parameters and residual"""
    with ctx_env({}, scratch_bytes=8 << 30) as c:
        c.enable_timing(True)
        a = run_batch(c, big["frames"], big["ns"], check_decode=False)
        assert c.launches[20] >= 1 and c.launches[18] >= 1, c.launches      # (k_fwd_loss's kind ran, behind the certified search: not k_last_layer)
    with ctx_env({"LINNE_AMD_FWD_LOSS_MW": "0"}, scratch_bytes=8 << 30) as c:
        b = run_batch(c, big["frames"], big["ns"], check_decode=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
    if "one_chunk" in big:
        o = big["one_chunk"]
        assert np.array_equal(a[0], o[0]) and np.array_equal(a[1], o[1]) and np.array_equal(a[2], o[2], equal_nan=True)


def test_last_layer_in_one_launch_with_eight_taps(ctx_env, oracle):
    """k_last_layer<8> (presets 2-4: layers 4 / 64 / 8): 3 328 stereo frames of 1024 samples = 26 624 jobs in one chunk, every frame
    with all four trials; against LINNE_AMD_LAST_LAYER=0 on every frame and against the oracle on a sample"""
    nch, bits, block, preset, F = 2, 16, 1024, 4, 3328
    base = np.stack([music(nch, block, bits, seed=700 + k) for k in range(64)])
    frames = np.ascontiguousarray(np.tile(base, ((F + 63) // 64, 1, 1))[:F])
    frames[1::2] = frames[1::2] // 3                           # (two loudness classes, so that the unit counts vary)
    ns = np.full(F, block, dtype=np.uint32)
    out = []
    for env in ({"LINNE_AMD_LAST_LAYER": "2"}, {"LINNE_AMD_LAST_LAYER": "0"}):
        with ctx_env(env, scratch_bytes=4 << 30) as c:
            shape = c.shape(nch, bits, block, preset, True)
            c.enable_timing(True)
            out.append(c.encode_frames_host(shape, frames, ns))
            assert c.last_launches(1) == 1, "expected one chunk"
            assert (c.last_launches(18) == 0) == (env["LINNE_AMD_LAST_LAYER"] == "2"), f"kind 18 launches: {c.last_launches(18)}"
    for a, b, what in zip(out[0], out[1], ("residual", "parameters", "statistics")):
        assert np.array_equal(a, b), what
    enc = oracle.encoder(nch, bits, 44100, block, preset, True)
    for f in list(range(0, 24)) + list(range(F - 8, F)):
        tap, ores = enc.hotpath(frames[f])
        _check_taps(tap, out[0][1][f], out[0][2][f], preset, nch, f"frame {f}")
        assert np.array_equal(ores, out[0][0][f]), f"frame {f}"
    enc.close()


@pytest.mark.parametrize("devices,group", [([0, 0], 0), ([0, 0, 0], 37), ([0], 100)])
def test_fan_out_over_several_contexts_in_one_process(oracle, big, devices, group):
    """LINNEAmd_MultiEncodeFramesHost / MultiDecodeFramesHost (direct per-GPU H2D/D2H): groups of frames go round-
    robin to per-device contexts, one host thread each.  A one-GPU box lists device 0 several times (separate contexts,
    arenas and streams): every frame must equal the single-context result and the oracle's, in the caller's order, with a
    ragged tail in the last group and group sizes that do not divide the batch"""
    F = 700
    frames, ns = big["frames"][-F:], big["ns"][-F:]
    m = linne_amd.Multi(devices, scratch_bytes=1 << 30)
    try:
        assert m.num_devices == len(devices)
        res, prm, st, plan = m.encode_frames_host(m_shape(), frames, ns, group_frames=group, want_plan=True)
        back = m.decode_frames_host(m_shape(), res, prm, ns, group_frames=group)
    finally:
        m.close()
    back[-1, :, TAIL:] = frames[-1, :, TAIL:]
    assert np.array_equal(back, frames)
    off = NBIG - F
    idx = sample_indices(F, seed=6, nmid=32)
    want = oracle_taps(oracle, big["frames"], big["ns"], [off + f for f in idx], big["cache"])
    compare_sample({f - off: v for f, v in want.items()}, ns, res, prm, st, f"fan-out {devices}")
    c = linne_amd.Context(0, scratch_bytes=2 << 30, use_torch_stream=False)
    try:
        r1 = run_batch(c, frames, ns, check_decode=False)
    finally:
        c.close()
    assert np.array_equal(r1[0], res) and np.array_equal(r1[1], prm) and np.array_equal(r1[2], st, equal_nan=True)
    # the Rice plan came along: packing with it gives the same blocks as the host's own search
    a, _ = linne_amd.pack_frames(m_shape(), frames[:8], res[:8], prm[:8], st[:8], ns[:8], 0.0, 2)
    b, _ = linne_amd.pack_frames(m_shape(), frames[:8], res[:8], prm[:8], st[:8], ns[:8], 0.0, 2, plan=plan[:8])
    assert a == b


def m_shape():
    return linne_amd.Shape(NCH, BITS, BLOCK, PRESET, 1)


@pytest.mark.parametrize("devices,group", [("0,0", "3"), ("0,0,0", "2")])
def test_whole_stream_api_over_several_devices(product, oracle, monkeypatch, devices, group):
    """LINNE_AMD_DEVICES=0,0: EncodeWhole / DecodeWhole rotate their frame groups over the listed devices (here two or three
    contexts on the one GPU); the .lnn bytes equal the oracle's -- Q2's state threads through the groups in stream order -- with
    SILENT / RAW blocks and a ragged tail in the stream, and decode restores the input"""
    from signals import waveform
    monkeypatch.setenv("LINNE_AMD_DEVICES", devices)
    monkeypatch.setenv("LINNE_AMD_GROUP", group)
    block = 2048
    parts = [music(2, 11 * block, 16, seed=31), np.zeros((2, 2 * block), dtype=np.int32), waveform("white_noise", 2, 3 * block, 16, seed=2),
             music(2, 7 * block + 555, 16, seed=32)]
    x = np.concatenate(parts, axis=1)
    mine = product.encode_whole(x, 16, 44100, block, 7, True)
    assert mine == oracle.encode_whole(x, 16, 44100, block, 7, True)
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)


@pytest.mark.parametrize("preset,nch,bits", [(0, 2, 16), (2, 1, 16), (4, 2, 24), (5, 8, 16), (7, 2, 16)])
def test_decode_throughput_forms_at_the_batch_sizes_that_pick_them(preset, nch, bits):
    """DecodeFramesDevice with nothing forced on batches of > 20 480 channel-frames (short blocks keep them cheap): k_synth_rows for
    the long layer of every preset family (32 / 64 / 128 taps), k_synth_rows8 for the layers of <= 16 taps, k_deemph_lr with and without
    the fused MS -> LR (1, 2 and 8 channels), several rounds of waves per SIMD, frames of a dozen lengths in the batch.  decode(encode(x))
    must be x, and what lies behind a frame's end must stay."""
    for v in ("LINNE_AMD_DECODE_KERNEL", "LINNE_AMD_DECODE_ROWS8"):
        assert v not in os.environ
    block = 1024
    F = 20480 // nch + 77
    base = np.stack([music(nch, block, bits, seed=100 * preset + k) for k in range(64)])             # [64][nch][block]
    frames = np.ascontiguousarray(np.tile(base, ((F + 63) // 64, 1, 1))[:F])
    rng = np.random.default_rng(5 + preset)
    ns = np.full(F, block, dtype=np.uint32)
    pool = np.array([1, 17, 130, 500, 777, block - 3, block // 2 + 1], dtype=np.uint32)
    where = rng.choice(F, size=200, replace=False)
    ns[where] = rng.choice(pool, size=200)
    ns[-1] = 777
    for f in np.flatnonzero(ns < block):
        frames[f, :, int(ns[f]):] = 0
    c = linne_amd.Context(0)
    try:
        shape = c.shape(nch, bits, block, preset, nch >= 2)
        res, prm, st = c.encode_frames_host(shape, frames, ns)
        marked = res.copy()
        for f in np.flatnonzero(ns < block):
            marked[f, :, int(ns[f]):] = -123456
        dec = c.decode_frames_host(shape, marked, prm, ns)
        full = ns == block
        if not np.array_equal(dec[full], frames[full]):
            # (seen once in some hundred runs and never again in 300 rounds of tools/poison_repro.py: say which side it was)
            wrong = [int(f) for f in np.flatnonzero(full) if not np.array_equal(dec[f], frames[f])]
            res2, prm2, _ = c.encode_frames_host(shape, frames, ns)
            dec2 = c.decode_frames_host(shape, marked, prm, ns)
            pytest.fail(f"decode(encode(x)) != x in {len(wrong)} full frames (first {wrong[:8]}); a second encode gives the same residual: "
                        f"{np.array_equal(res2, res)}, the same parameters: {np.array_equal(prm2, prm)}; a second decode of the first encode's output "
                        f"gives the same PCM: {np.array_equal(dec2, dec)}, the input: {np.array_equal(dec2[full], frames[full])}")
    finally:
        c.close()
    full = ns == block
    assert np.array_equal(dec[full], frames[full])
    for f in np.flatnonzero(~full):
        n = int(ns[f])
        assert np.array_equal(dec[f, :, :n], frames[f, :, :n]), f"frame {f} (n = {n})"
        assert np.all(dec[f, :, n:] == -123456), f"frame {f}: samples behind its end were written"


def test_the_default_two_streams_give_the_one_stream_result(monkeypatch):
    """a call whose halves keep the large-batch kernel forms (>= 32 768 jobs each) is cut in two over two compute streams by default
    (lnn_device.hip, the rule at the chunk loop); residual, parameters and statistics must be those of LINNE_AMD_STREAMS=1, bit for bit
    (short blocks keep the batch cheap: 8 320 stereo frames of 1024 samples, a dozen frame lengths among them)"""
    assert "LINNE_AMD_STREAMS" not in os.environ
    nch, bits, block, preset, F = 2, 16, 1024, 7, 8320
    base = np.stack([music(nch, block, bits, seed=900 + k) for k in range(64)])
    frames = np.ascontiguousarray(np.tile(base, ((F + 63) // 64, 1, 1))[:F])
    rng = np.random.default_rng(77)
    ns = np.full(F, block, dtype=np.uint32)
    where = rng.choice(F, size=300, replace=False)
    ns[where] = rng.choice(np.array([130, 500, 777, block - 3, block // 2 + 1], dtype=np.uint32), size=300)
    for f in np.flatnonzero(ns < block):
        frames[f, :, int(ns[f]):] = 0
    out, chunks = [], []
    for streams in (None, "1"):
        if streams:
            monkeypatch.setenv("LINNE_AMD_STREAMS", streams)
        c = linne_amd.Context(0)
        try:
            shape = c.shape(nch, bits, block, preset, True)
            c.enable_timing(True)
            out.append(c.encode_frames_host(shape, frames, ns))
            chunks.append(c.last_launches(1))       # k_prep spans = chunks of the call
        finally:
            c.close()
    assert chunks == [2, 1], f"the default call was not cut in two (chunks per call: {chunks})"
    for a, b, what in zip(out[0], out[1], ("residual", "parameters", "statistics")):
        assert np.array_equal(a, b), what


def test_the_default_split_at_the_real_frame_size_against_the_oracle(oracle):
    """What bench.py's timed batch runs, in small: 8 192 stereo frames of 10 240 samples at -m 7 (65 536 jobs) with NOTHING forced, an
    arena that holds each half in one chunk (as bench.py's does).  The call must have been cut in two over the two compute streams
    (the rule at the chunk loop of LINNEAmd_EncodeFramesDevice), every large-batch kernel form must have run, a sample of the output --
    the first 64 frames, the last 64 with the ragged tail, 128 random ones from both halves -- must equal the oracle's bit for bit
    (libs/linne_encoder/src/linne_encoder.c:594-752), and the whole batch must decode to the input."""
    for v in ("LINNE_AMD_STREAMS", "LINNE_AMD_HIST", "LINNE_AMD_FWD_LOSS", "LINNE_AMD_SEARCH_LONG", "LINNE_AMD_DECODE_KERNEL"):
        assert v not in os.environ
    F = 8192
    x = music(NCH, (F - 1) * BLOCK + TAIL, BITS, seed=404)
    flat = np.zeros((NCH, F * BLOCK), dtype=np.int32)
    flat[:, :x.shape[1]] = x
    del x
    frames = np.ascontiguousarray(flat.reshape(NCH, F, BLOCK).transpose(1, 0, 2))
    del flat
    ns = np.full(F, BLOCK, dtype=np.uint32)
    ns[-1] = TAIL
    c = linne_amd.Context(0, scratch_bytes=16 << 30)
    try:
        c.enable_timing(True)
        res, prm, st = run_batch(c, frames, ns)
        assert c.launches[1] == 2, f"chunks of the call: {c.launches[1]} (expected the two halves of the default split)"
        assert kinds_that_ran(c, (20, 21, 22, 23)) == {20, 21, 22, 23}, c.launches
        assert c.last_fallback_count() >= 0
    finally:
        c.close()
    idx = sample_indices(F, seed=11)
    assert any(f < F // 2 for f in idx[64:-64]) and any(f >= F // 2 for f in idx[64:-64])      # both halves sampled in the middle
    compare_sample(oracle_taps(oracle, frames, ns, idx), ns, res, prm, st, "default split")


def test_bench_two_rank_rehearsal_line_is_complete(tmp_path):
    """`python bench.py --gpus 2` on the one-GPU box (both ranks on device 0, transfers over gloo: BENCH_BACKEND=gloo -- a rehearsal of
    the N > 1 path, not an RCCL measurement).  The line an 8-GPU run would be judged by must be complete: the reference's CPU encoder
    and decoder timed in the SAME run (north_star; libs/linne_encoder/src/linne_encoder.c:774-862, libs/linne_decoder/src/linne_decoder.c:564-668),
    parity on every rank, the N = 1 figures of the transport legs and the ratios over them."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--minutes", "4", "--steps", "2", "--cpu-frames-per-thread", "8", "--scratch-gib", "8"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["scaling"] == "weak"
    assert line["encode_sample_parity"] is True and line["decode_bit_exact"] is True
    cpu, cpud = line["cpu_baseline"], line["cpu_baseline_decode"]
    assert cpu and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["kind"] in ("reference", "port")
    if cpu["kind"] == "reference":
        assert cpud and cpud["value"] > 0 and cpud["bit_exact"] is True
    assert line["hot_path_resident_over_cpu_encodeblock"] > 1
    n1 = line["n1_reference"]
    assert n1["resident_frames_per_s"] > 0 and n1["direct_h2d_frames_per_s"] > 0 and line["value_over_n1"] > 0
    tr = line["transports"]
    assert tr["direct_h2d"]["over_n1"] > 0 and tr["direct_h2d"]["results_equal_resident_path"] is True
    ex = tr["gloo_scatter_gather_rehearsal"]
    assert ex["over_n1"] > 0 and ex["results_equal_resident_path"] is True and ex["ranks"] == 2
    assert line["rccl_ranks"] is None                 # gloo moved the bytes: nothing may claim RCCL counted ranks
    assert "legs_timed_out" not in line
    assert line["roofline"]["frac"] > 0 and line["roofline"]["one_stream_step_ms"] > 0
