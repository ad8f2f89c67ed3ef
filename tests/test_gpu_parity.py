"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the oracle on the same inputs.
Bar: bit-exact (all outputs are integers; the double-precision analysis feeds an 8-bit quantiser and an argmin,
so any deviation in operation order would show up as differing coefficients / unit counts)."""
import os
import numpy as np
import pytest

import linne_amd
from refs import fnv1a64
from signals import WAVEFORMS, music, music_frames, waveform

pytestmark = pytest.mark.gpu


def _check_taps(tap, rec, stats, preset, nch, where):
    for ch in range(nch):
        t, r = tap.ch[ch], rec[ch]
        assert list(t.preem_prev) == list(r[0:2]), f"{where} ch{ch}: pre-emphasis prev"
        assert list(t.preem_coef) == list(r[2:4]), f"{where} ch{ch}: pre-emphasis coef"
        nl = len(linne_amd.PRESET_LAYERS[preset])
        assert list(t.num_units)[:nl] == list(r[4:4 + nl]), f"{where} ch{ch}: units {list(t.num_units)} vs {list(r[4:7])}"
        assert list(t.rshift)[:nl] == list(r[7:7 + nl]), f"{where} ch{ch}: rshift"
        off = 10
        for l, P in enumerate(linne_amd.PRESET_LAYERS[preset]):
            assert list(t.coef[l][:P]) == list(r[off:off + P]), f"{where} ch{ch} layer{l}: coefficients"
            off += P
        st = stats[ch]
        assert st[linne_amd.ST_R0] == t.est_r0 or (np.isnan(st[linne_amd.ST_R0]) and np.isnan(t.est_r0)), f"{where} ch{ch}: SIN-window r0 {st[0]!r} vs {t.est_r0!r}"      # (a one-sample frame: 0 / 0 in the reference's window too)
        assert int(st[linne_amd.ST_BEST]) == t.best_pass, f"{where} ch{ch}: best regulariser"
        assert st[linne_amd.ST_LOSS] == t.pass_loss[t.best_pass], f"{where} ch{ch}: L1 loss"
        assert st[linne_amd.ST_TAIL] == t.parcor_tail, f"{where} ch{ch}: parcor tail (Q2)"


def test_smoke():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.parametrize("nch,bits,block,preset,ms,nframes", [
    (2, 16, 10240, 7, True, 3),
    (1, 16, 10240, 4, False, 2),
    (2, 16, 1024, 0, True, 4),
    (2, 24, 4096, 5, True, 2),
    (8, 24, 10240, 7, True, 1),
    (3, 8, 1024, 2, False, 3),
    (2, 16, 2048, 6, False, 2),
])
def test_hotpath_full_frames(ctx, oracle, nch, bits, block, preset, ms, nframes):
    frames = music_frames(nframes, nch, block, bits, seed=block + preset)
    shape = ctx.shape(nch, bits, block, preset, ms)
    res, prm, st = ctx.encode_frames_host(shape, frames)
    for f in range(nframes):
        enc = oracle.encoder(nch, bits, 44100, block, preset, ms)
        tap, ores = enc.hotpath(frames[f])
        enc.close()
        _check_taps(tap, prm[f], st[f], preset, nch, f"frame {f}")
        assert np.array_equal(ores, res[f]), f"frame {f}: residual"
    # decode hot path: inverse of the encode one
    dec = ctx.decode_frames_host(shape, res, prm)
    assert np.array_equal(dec, frames)


@pytest.mark.parametrize("n", [129, 680, 2000, 9280, 3001, 8, 1000, 10239])
def test_hotpath_ragged_tail_frames(ctx, oracle, n):
    """tail frames: zero padding to the analysis length and the odd-window quirk Q1 (n = 680, 2000, 9280 give odd
    sub-lengths 85, 125, 145)"""
    nch, bits, block, preset = 2, 16, 10240, 7
    x = music(nch, block + n, bits, seed=n)
    frames = np.zeros((2, nch, block), dtype=np.int32)
    frames[0] = x[:, :block]
    frames[1, :, :n] = x[:, block:]
    ns = np.array([block, n], dtype=np.uint32)
    shape = ctx.shape(nch, bits, block, preset, True)
    res, prm, st = ctx.encode_frames_host(shape, frames, ns)
    enc = oracle.encoder(nch, bits, 44100, block, preset, True)
    for f in range(2):
        tap, ores = enc.hotpath(frames[f][:, :ns[f]])
        _check_taps(tap, prm[f], st[f], preset, nch, f"frame {f} n={ns[f]}")
        assert np.array_equal(ores, res[f][:, :ns[f]])
    enc.close()
    dec = ctx.decode_frames_host(shape, res, prm, ns)
    assert np.array_equal(dec[0], frames[0]) and np.array_equal(dec[1][:, :n], frames[1][:, :n])


@pytest.mark.parametrize("kind", WAVEFORMS)
@pytest.mark.parametrize("nch,bits,preset", [(1, 16, 0), (2, 16, 7), (8, 8, 4), (2, 24, 7)])
def test_lnn_bytes_reference_waveforms(product, oracle, kind, nch, bits, preset):
    """the reference's round-trip matrix (test/linne_encode_decode/main.cpp:335-536), pinned at the byte level:
    EncodeWhole through the drop-in API == the oracle's stream; DecodeWhole restores the input"""
    x = waveform(kind, nch, 8192, bits, seed=nch * 100 + bits)
    ms = nch >= 2
    mine = product.encode_whole(x, bits, 8000, 1024, preset, ms)
    want = oracle.encode_whole(x, bits, 8000, 1024, preset, ms)
    assert len(mine) == len(want) and mine == want, f"{kind}: .lnn differs ({fnv1a64(mine)} vs {fnv1a64(want)})"
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec[:nch, :8192], x)


def test_lnn_bytes_music_with_tail(product, oracle, reference):
    x = music(2, 3 * 10240 + 2000, 16, seed=11)
    mine = product.encode_whole(x, 16, 44100, 10240, 7, True)
    assert mine == oracle.encode_whole(x, 16, 44100, 10240, 7, True)
    assert mine == reference.encode_whole(x, 16, 44100, 10240, 7, True)
    blocks = product.encode_blocks(x, 16, 44100, 10240, 7, True)          # the CLI's block-at-a-time loop
    assert blocks == mine
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)
    ret, dec = reference.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)


def test_decode_errors(product, oracle):
    """linne_decoder_test.cpp:470-579: sync corruption, payload corruption, truncation"""
    x = music(2, 4096, 16, seed=5)
    good = oracle.encode_whole(x, 16, 44100, 2048, 4, True)
    bad = bytearray(good); bad[30] ^= 0xFF
    assert product.decode_whole(bytes(bad))[0] == 2            # INVALID_FORMAT
    bad = bytearray(good); bad[60] ^= 0xFF
    assert product.decode_whole(bytes(bad))[0] == 6            # DETECT_DATA_CORRUPTION
    assert product.decode_whole(good[:len(good) - 7])[0] == 4  # INSUFFICIENT_DATA


def test_full_size_round_trip_property(ctx):
    """size-independent property at a larger batch: decode(encode(x)) == x for 64 stereo frames"""
    import torch
    frames = music_frames(64, 2, 10240, 16, seed=99)
    shape = ctx.shape(2, 16, 10240, 7, True)
    res, prm, st = ctx.encode_frames_host(shape, frames)
    assert np.array_equal(ctx.decode_frames_host(shape, res, prm), frames)
    assert (prm[:, :, linne_amd.PRM_UNITS:linne_amd.PRM_UNITS + 3] >= 1).all()
    assert ctx.last_fallback_count() == 0        # ordinary audio: every search is certified by the order-free sums


def test_golden_streams_through_the_drop_in_api(product):
    """committed golden vectors (generated by the real reference, tests/golden/make_golden.py)"""
    import hashlib
    import json
    import os
    from test_oracle_cpu import GOLD, golden_small, read_wav
    for i, x, bits, rate, block, preset, ms, want in golden_small():
        got = product.encode_whole(x, bits, rate, block, preset, ms)
        assert got == want, f"small case {i}"
        ret, dec = product.decode_whole(want)
        assert ret == 0 and np.array_equal(dec[:x.shape[0], :x.shape[1]], x)
    h = json.load(open(os.path.join(GOLD, "golden_hashes.json")))
    for name, e in h.items():
        if name.startswith("large/"):
            x = music(*e["music_args"])
            if hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() != e["input_sha256"]:
                continue
            got = product.encode_whole(x, e["music_args"][2], e["rate"], e["block"], e["preset"], bool(e["ms"]))
        elif name.startswith("wav/"):
            x, rate, bits = read_wav(os.path.join(GOLD, name[4:].rsplit("_m", 1)[0]))
            got = product.encode_whole(x, bits, rate, e["block"], e["preset"], bool(e["ms"]))
        else:
            continue
        assert len(got) == e["bytes"] and hashlib.sha256(got).hexdigest() == e["sha256"], name
        ret, dec = product.decode_whole(got)
        assert ret == 0 and np.array_equal(dec, x), name


def test_reference_cli_links_against_liblinne_amd_unchanged(tmp_path):
    """oracle/_ref/linne_dropin = the reference's tools/linne_codec/linne_codec.c compiled against this repo's
    include/ and linked with liblinne_amd.so; its .lnn must equal the reference CLI's, and decode must restore the WAV"""
    import os
    import subprocess
    from refs import ROOT
    dropin, refcli = os.path.join(ROOT, "oracle", "_ref", "linne_dropin"), os.path.join(ROOT, "oracle", "_ref", "linne_ref")
    if not (os.path.exists(dropin) and os.path.exists(refcli)):
        pytest.skip("oracle/_ref CLI binaries not built")
    wav = os.path.join(ROOT, "tests", "golden", "ref_16bit_2ch.wav")
    a, b, w = str(tmp_path / "a.lnn"), str(tmp_path / "b.lnn"), str(tmp_path / "back.wav")
    subprocess.run([refcli, "-e", "-m", "7", wav, a], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([dropin, "-e", "-m", "7", wav, b], check=True, stdout=subprocess.DEVNULL)
    assert open(a, "rb").read() == open(b, "rb").read()
    subprocess.run([dropin, "-d", b, w], check=True, stdout=subprocess.DEVNULL)
    assert open(w, "rb").read()[-176400:] == open(wav, "rb").read()[-176400:]     # PCM payload (44100 x 2 ch x 2 B)


@pytest.mark.parametrize("kind", ["silence", "positive_const", "negative_const", "nyquist", "sine", "chirp", "white_noise"])
def test_hotpath_degenerate_signals(ctx, oracle, kind):
    """signals whose trial losses tie exactly (all-zero residuals) or nearly: the certified order-free search must
    hand them to the exact ordered chains and still pick the reference's unit counts"""
    nch, bits, block, preset = 2, 16, 1024, 7
    x = waveform(kind, nch, 4 * block, bits, seed=3)
    frames = np.ascontiguousarray(x.reshape(nch, 4, block).transpose(1, 0, 2))
    shape = ctx.shape(nch, bits, block, preset, True)
    res, prm, st = ctx.encode_frames_host(shape, frames)
    for f in range(4):
        enc = oracle.encoder(nch, bits, 44100, block, preset, True)
        tap, ores = enc.hotpath(frames[f])
        enc.close()
        _check_taps(tap, prm[f], st[f], preset, nch, f"{kind} frame {f}")
        assert np.array_equal(ores, res[f])
    if kind in ("silence", "positive_const", "negative_const"):
        assert ctx.last_fallback_count() > 0, "exact ties must take the ordered-chain fallback"


@pytest.mark.parametrize("group", ["2", "5"])
def test_whole_stream_pipeline_over_staging_slots(product, oracle, monkeypatch, group):
    """EncodeWhole / DecodeWhole rotate groups of frames over three staging slots (pinned host + device buffers, copy
    streams beside the kernel stream): with tiny groups a short stream exercises slot reuse, a ragged tail, and
    SILENT / RAW blocks between COMPRESS ones; the bytes must not depend on the grouping"""
    monkeypatch.setenv("LINNE_AMD_GROUP", group)
    block = 2048
    parts = [music(2, 9 * block, 16, seed=21), np.zeros((2, 2 * block), dtype=np.int32), waveform("white_noise", 2, 3 * block, 16, seed=2),
             music(2, 5 * block + 777, 16, seed=22)]
    x = np.concatenate(parts, axis=1)
    mine = product.encode_whole(x, 16, 44100, block, 7, True)
    want = oracle.encode_whole(x, 16, 44100, block, 7, True)
    assert mine == want
    types, off = set(), 30
    while off < len(mine):
        types.add(mine[off + 8]); off += int.from_bytes(mine[off + 2:off + 6], "big") + 6
    assert types == {0, 1, 2}                                   # compress, silent and raw all occur
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)
    # a stream cut in the middle of a later group: the error is reported and the blocks before it are delivered
    cut = product.decode_whole(mine[:len(mine) * 2 // 3])
    assert cut[0] == 4                                          # INSUFFICIENT_DATA
    assert np.array_equal(cut[1][:, :6 * block], x[:, :6 * block])


@pytest.mark.parametrize("nch,bits,block,preset,tail", [(2, 16, 10240, 7, 9280), (2, 16, 2048, 4, 777), (1, 24, 4096, 0, 4096), (3, 8, 1024, 2, 1000)])
def test_device_rice_plan_equals_host_search(ctx, nch, bits, block, preset, tail):
    """partition means, parameters and the partition-order search of the Rice coder on the device.
    The plan must reproduce what the host's own search writes: packing with the plan == packing without it, byte for
    byte, on music, noise and small/large residuals, full and ragged frames."""
    import torch
    ms = nch >= 2
    F = 6
    frames = music_frames(F, nch, block, bits, seed=31 + nch)
    frames[2] = waveform("white_noise", nch, block, bits, seed=4)
    frames[3] //= 64                                             # small residuals: low parameters
    ns = np.full(F, block, dtype=np.uint32); ns[-1] = tail
    frames[-1, :, tail:] = 0
    shape = ctx.shape(nch, bits, block, preset, ms)
    pcm = torch.from_numpy(frames).cuda()
    res, prm, st = ctx.encode_frames(shape, pcm, ns)
    plan = ctx.rice_plan(shape, res, ns)
    ctx.synchronize()
    res, prm, st, plan = res.cpu().numpy(), prm.cpu().numpy(), st.cpu().numpy(), plan.cpu().numpy()
    a, _ = linne_amd.pack_frames(shape, frames, res, prm, st, ns, 0.0, 2)
    b, _ = linne_amd.pack_frames(shape, frames, res, prm, st, ns, 0.0, 2, plan=plan)
    assert a == b
    assert (plan[:, :, 1] == 0).all()                            # no mean of these signals sits on a parameter step
    assert (plan[:, :, 0] <= 10).all()
    # a plan whose flag is raised is ignored (the host searches itself): same bytes again
    plan2 = plan.copy(); plan2[:, :, 1] = 1; plan2[:, :, 16:] = 0
    c, _ = linne_amd.pack_frames(shape, frames, res, prm, st, ns, 0.0, 2, plan=plan2)
    assert a == c


def test_own_cli_matches_the_reference_cli(tmp_path):
    """tools/cli (own WAV reader/writer, own option parsing) on liblinne_amd.so: the .lnn it writes -- whole-stream and
    block-at-a-time -- equals the reference CLI's byte for byte, decoding gives the reference's WAV, and --batch runs
    several files through one handle (BASELINE config 4's many-track use)"""
    import os
    import shutil
    import subprocess
    from refs import ROOT
    cli, refcli = os.path.join(ROOT, "linne_amd", "linne_amd_cli"), os.path.join(ROOT, "oracle", "_ref", "linne_ref")
    if not (os.path.exists(cli) and os.path.exists(refcli)):
        pytest.skip("CLI binaries not built")
    wav = os.path.join(ROOT, "tests", "golden", "ref_16bit_2ch.wav")
    wav2 = os.path.join(ROOT, "tests", "golden", "ref_a.wav")
    ref_lnn, mine, mine_b = str(tmp_path / "ref.lnn"), str(tmp_path / "mine.lnn"), str(tmp_path / "mine_b.lnn")
    subprocess.run([refcli, "-e", "-m", "7", wav, ref_lnn], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([cli, "-e", "-m", "7", wav, mine], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([cli, "-e", "--mode=7", "-B", "-q", wav, mine_b], check=True, stdout=subprocess.DEVNULL)
    want = open(ref_lnn, "rb").read()
    assert open(mine, "rb").read() == want and open(mine_b, "rb").read() == want
    ref_wav, my_wav = str(tmp_path / "ref.wav"), str(tmp_path / "mine.wav")
    subprocess.run([refcli, "-d", ref_lnn, ref_wav], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([cli, "-d", mine, my_wav], check=True, stdout=subprocess.DEVNULL)
    assert open(my_wav, "rb").read() == open(ref_wav, "rb").read()
    # batch: two different files (different formats) through one encoder handle, then one decoder handle
    out = tmp_path / "batch"; out.mkdir()
    shutil.copy(wav, tmp_path / "one.wav"); shutil.copy(wav2, tmp_path / "two.wav")
    subprocess.run([cli, "-e", "-m", "5", "-q", "--batch", str(out), str(tmp_path / "one.wav"), str(tmp_path / "two.wav")], check=True)
    for name, src in (("one", wav), ("two", wav2)):
        r = str(tmp_path / (name + "_ref.lnn"))
        subprocess.run([refcli, "-e", "-m", "5", src, r], check=True, stdout=subprocess.DEVNULL)
        assert (out / (name + ".lnn")).read_bytes() == open(r, "rb").read(), name
    back = tmp_path / "back"; back.mkdir()
    subprocess.run([cli, "-d", "-q", "--batch", str(back), str(out / "one.lnn"), str(out / "two.lnn")], check=True)
    for name in ("one", "two"):
        r = str(tmp_path / (name + "_ref.wav"))
        subprocess.run([refcli, "-d", str(out / (name + ".lnn")), r], check=True, stdout=subprocess.DEVNULL)
        assert (back / (name + ".wav")).read_bytes() == open(r, "rb").read(), name


def _params_from_tap(tap, preset, nch):
    """the oracle's per-channel parameters as a [C][PARAM_WORDS] record of include/linne_amd.h (pre-emphasis prev x2, coef x2,
    units, shifts, then the layers' coefficients in filter order)"""
    rec = np.zeros((nch, linne_amd.PARAM_WORDS), dtype=np.int32)
    for ch in range(nch):
        t = tap.ch[ch]
        rec[ch, 0:2] = list(t.preem_prev); rec[ch, 2:4] = list(t.preem_coef)
        L = linne_amd.PRESET_LAYERS[preset]
        rec[ch, 4:4 + len(L)] = list(t.num_units)[:len(L)]
        rec[ch, 7:7 + len(L)] = list(t.rshift)[:len(L)]
        off = 10
        for l, P in enumerate(L):
            rec[ch, off:off + P] = list(t.coef[l][:P])
            off += P
    return rec


def _force_decode_form(monkeypatch, kernel):
    """LINNE_AMD_DECODE_KERNEL for a test; `rows` / `rows4`: the throughput form with EIGHT / FOUR channel-frames per wave in the layers of
    <= 16 taps (the batch-size rule would pick four for batches as small as a test's), layer 0 + de-emphasis + MS -> LR in one launch
    (k_synth_l0_de); `rows_nf`: the same with those three as launches of their own"""
    monkeypatch.setenv("LINNE_AMD_DECODE_KERNEL", "rows" if kernel in ("rows", "rows4", "rows_nf") else kernel)
    if kernel in ("rows", "rows4", "rows_nf"):
        monkeypatch.setenv("LINNE_AMD_DECODE_ROWS8", "0" if kernel == "rows4" else "1")
    if kernel == "rows_nf":         # layer 0, the de-emphasis and MS -> LR as launches of their own (k_synth_rows8 + k_deemph_lr) instead of k_synth_l0_de
        monkeypatch.setenv("LINNE_AMD_DECODE_FUSED", "0")


@pytest.mark.parametrize("kernel", ["wave", "lanes", "pipe", "rows", "rows4", "rows_nf"])
@pytest.mark.parametrize("nch,bits,block,preset,tail", [(2, 16, 10240, 7, 9280), (2, 16, 2048, 4, 777), (1, 16, 1024, 0, 130), (8, 24, 4096, 7, 4096), (3, 8, 1024, 2, 1000), (2, 16, 4096, 5, 3001)])
def test_decode_kernels_agree(ctx, oracle, monkeypatch, kernel, nch, bits, block, preset, tail):
    """DecodeFramesDevice picks its kernels by batch size (a wave per stage of the cascade for small batches; four / eight
    channel-frames per wave, or lanes = channel-frames, for large ones).  Every form is fed what the ORACLE's encoder wrote
    (its residual and parameters, not the product's) and must give the oracle's own synthesis of it
    (oracle_decode_frame_hotpath: libs/linne_decoder/src/linne_decoder.c:503-522, linne_lpc_synthesize.c:8-83) -- which is the input --
    on every preset family, ragged tails included; then the product's own encode output must decode to the input as well.
    (`rows4`: the throughput form with four channel-frames per wave for the short layers too; `rows` takes eight there.)"""
    _force_decode_form(monkeypatch, kernel)
    ms = nch >= 2
    F = 5
    frames = music_frames(F, nch, block, bits, seed=77 + nch + preset)
    frames[1] = waveform("chirp", nch, block, bits, seed=3)
    ns = np.full(F, block, dtype=np.uint32); ns[-1] = tail
    frames[-1, :, tail:] = 0
    shape = ctx.shape(nch, bits, block, preset, ms)
    ores = np.zeros_like(frames)
    oprm = np.zeros((F, nch, linne_amd.PARAM_WORDS), dtype=np.int32)
    want = np.zeros_like(frames)
    for f in range(F):
        n = int(ns[f])
        enc = oracle.encoder(nch, bits, 44100, block, preset, ms)
        tap, r = enc.hotpath(frames[f][:, :n])
        enc.close()
        ores[f, :, :n] = r
        oprm[f] = _params_from_tap(tap, preset, nch)
        want[f, :, :n] = oracle.decode_hotpath([tap.ch[ch] for ch in range(nch)], r, bits, block, preset, ms)
        assert np.array_equal(want[f, :, :n], frames[f, :, :n]), f"frame {f}: the oracle's own round trip"
    dec = ctx.decode_frames_host(shape, ores, oprm, ns)
    for f in range(F):
        n = int(ns[f])
        assert np.array_equal(dec[f, :, :n], want[f, :, :n]), f"frame {f}: HIP synthesis of the oracle's residual and parameters"
    res, prm, st = ctx.encode_frames_host(shape, frames, ns)
    dec = ctx.decode_frames_host(shape, res, prm, ns)
    for f in range(F):
        n = int(ns[f])
        assert np.array_equal(dec[f, :, :n], frames[f, :, :n]), f"frame {f}"


@pytest.mark.parametrize("kernel", ["rows", "rows4", "rows_nf", "lanes", "pipe"])
@pytest.mark.parametrize("nch,bits,block,preset,F", [(2, 16, 2048, 7, 37), (3, 24, 1024, 4, 23), (1, 16, 4096, 5, 70), (8, 16, 1024, 7, 9)])
def test_decode_forms_with_frames_of_many_lengths_in_one_batch(ctx, monkeypatch, kernel, nch, bits, block, preset, F):
    """a batch as many tracks back to back make it: frames of a dozen lengths in any order (so the four channel-frames of a
    k_synth_rows wave, the 64 rows of a k_deemph_lr / k_synth_small block and the rows of its last, partial block end in
    different places, unit boundaries fall inside the 16-sample blocks, and the shortest frames are shorter than a layer's
    order); the decode must restore every frame's own samples and leave what lies behind them alone"""
    _force_decode_form(monkeypatch, kernel)
    rng = np.random.default_rng(99 + F)
    ms = nch >= 2
    frames = music_frames(F, nch, block, bits, seed=5 + nch)
    pool = np.concatenate([[block, 1, 129, block - 3, 17, block // 2 + 1], rng.integers(1, block + 1, size=6)])     # (an encode call takes up to 16 distinct lengths)
    ns = rng.choice(pool, size=F).astype(np.uint32)
    ns[:6] = pool[:6]
    for f in range(F):
        frames[f, :, int(ns[f]):] = 0
    shape = ctx.shape(nch, bits, block, preset, ms)
    res, prm, st = ctx.encode_frames_host(shape, frames, ns)
    marked = res.copy()
    for f in range(F):
        marked[f, :, int(ns[f]):] = -123456                      # what lies behind a frame's end is nobody's data: it must stay
    dec = ctx.decode_frames_host(shape, marked, prm, ns)
    for f in range(F):
        n = int(ns[f])
        assert np.array_equal(dec[f, :, :n], frames[f, :, :n]), f"frame {f} (n = {n})"
        assert np.all(dec[f, :, n:] == -123456), f"frame {f}: samples behind its end were written"


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("LINNE_FUZZ_SEEDS", "16"))))
def test_random_configurations_match_the_oracle(product, oracle, monkeypatch, seed):
    """randomised sweep over what the API accepts: channels, bit depth, preset, block size (even or odd), MS on/off, stream length
    with a ragged tail, and material that mixes music, silence, noise and a constant; the .lnn must equal the oracle's
    byte for byte and decode back to the input.  Odd seeds force the large-batch kernel forms (picked by batch size
    otherwise, and these streams are short) so that the sweep covers both."""
    if seed % 2:
        monkeypatch.setenv("LINNE_AMD_HIST", "1")
        monkeypatch.setenv("LINNE_AMD_FWD_LOSS", "1")
        monkeypatch.setenv("LINNE_AMD_STATS_ROWS", "1")          # the batch form of the block-type statistics (round 3)
    if seed % 3 == 2:
        monkeypatch.setenv("LINNE_AMD_DECODE_KERNEL", "lanes")   # (the default for these short streams is the pipelined latency form)
    if seed % 3 == 1:
        _force_decode_form(monkeypatch, "rows" if seed % 2 else "rows4")     # the batch form of the synthesis: eight / four channel-frames per wave in the short layers
    if seed % 4 == 3:
        monkeypatch.setenv("LINNE_AMD_PREP_GENERAL", "1")
    rng = np.random.default_rng(1000 + seed)
    nch = int(rng.integers(1, 9))
    bits = int(rng.choice([8, 16, 24]))
    preset = int(rng.integers(0, 8))
    maxp = 32 if preset < 2 else (64 if preset < 5 else 128)
    block = int(rng.choice([2 * int(rng.integers(max(maxp, 128) // 2 + 1, 1500)), 2 * int(rng.integers(max(maxp, 128) // 2 + 1, 1500)) + 1, 1024, 2048, 4096]))     # even and odd; the handle is created for 128-tap layers
    ms = bool(nch >= 2 and rng.integers(0, 2))
    nblocks = int(rng.integers(2, 6))
    total = nblocks * block + int(rng.integers(1, block))
    parts, left = [], total
    while left > 0:
        n = int(min(left, rng.integers(block // 2, 2 * block)))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            parts.append(np.zeros((nch, n), dtype=np.int32))
        elif kind == 1:
            parts.append(waveform("white_noise", nch, n, bits, seed=int(rng.integers(1 << 30))))
        elif kind == 2:
            parts.append(waveform("positive_const", nch, n, bits, seed=1))
        else:
            parts.append(music(nch, n, bits, seed=int(rng.integers(1 << 30))))
        left -= n
    x = np.concatenate(parts, axis=1)
    mine = product.encode_whole(x, bits, 44100, block, preset, ms)
    want = oracle.encode_whole(x, bits, 44100, block, preset, ms)
    assert mine == want, f"nch={nch} bits={bits} preset={preset} block={block} ms={ms} total={total}"
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)


@pytest.mark.parametrize("kernel", ["wave", "lanes", "pipe", "rows", "rows4", "rows_nf"])
def test_corrupt_streams_do_not_hang_or_fault(product, oracle, monkeypatch, kernel):
    """with the CRC check off a damaged payload reaches the parser and the GPU with arbitrary parameters (unit counts,
    shifts, coefficients, residuals): decoding must come back with a result code (and the device must stay usable)"""
    _force_decode_form(monkeypatch, kernel)
    rng = np.random.default_rng(4242)
    x = music(2, 6 * 2048 + 100, 16, seed=8)
    good = oracle.encode_whole(x, 16, 44100, 2048, 7, True)
    for trial in range(60):
        bad = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(41, len(bad)))                 # past the stream header and the first block header
            bad[pos] = int(rng.integers(0, 256))
        ret, dec = product.decode_whole(bytes(bad), check_crc=0)
        assert ret in range(8)
    ret, dec = product.decode_whole(good)                          # the device still works
    assert ret == 0 and np.array_equal(dec, x)


@pytest.mark.parametrize("speculate", ["0", "1"])
def test_search_with_and_without_the_fused_forward(product, oracle, monkeypatch, speculate):
    """by default the unit-count search of the first layers also writes the forward output of the one-unit trial, and the
    forward pass skips the jobs that chose one unit (LINNE_AMD_SPECULATE); both schedules must give the oracle's bytes --
    on music (mostly one unit) and on a signal whose statistics change inside the frame (several units)"""
    monkeypatch.setenv("LINNE_AMD_SPECULATE", speculate)
    block = 4096
    seg = [music(2, 3 * block, 16, seed=51)]
    rng = np.random.default_rng(7)
    t = np.arange(2 * block)
    burst = (8000 * np.sin(2 * np.pi * t * (0.01 + 0.2 * (t // 512 % 2))) * (t // 256 % 2) + rng.integers(-30, 30, size=2 * block)).astype(np.int32)
    seg.append(np.stack([burst, -burst // 2]))
    x = np.concatenate(seg, axis=1)
    mine = product.encode_whole(x, 16, 44100, block, 7, True)
    assert mine == oracle.encode_whole(x, 16, 44100, block, 7, True)
    # some channel-frame of the second part must have chosen more than one unit in layer 0 or 1 (else the test shows nothing)
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)


@pytest.mark.parametrize("products", ["0", "1"])
@pytest.mark.parametrize("preset", [1, 7])
def test_layer0_autocorrelation_forms_agree(product, oracle, monkeypatch, products, preset):
    """layer 0's lags come from k_autocorr_l0 (products staged in LDS, one lane per chain) for small batches and from
    k_autocorr_lane for large ones; LINNE_AMD_L0_PRODUCTS=0 forces the latter here: same bytes either way"""
    monkeypatch.setenv("LINNE_AMD_L0_PRODUCTS", products)
    x = music(2, 5 * 4096 + 1234, 16, seed=61 + preset)
    assert product.encode_whole(x, 16, 44100, 4096, preset, True) == oracle.encode_whole(x, 16, 44100, 4096, preset, True)


@pytest.mark.parametrize("fwd_loss,lev_ride", [("0", "1"), ("1", "0"), ("1", "1")])
@pytest.mark.parametrize("preset", [3, 7])
def test_last_layer_loss_kernel_and_riding_levinson_trials(product, oracle, monkeypatch, fwd_loss, lev_ride, preset):
    """the last layer's forward pass + ordered loss come from one kernel (k_fwd_loss) for frames whose unit lengths are all
    multiples of 4 and from k_fir2<1> + k_chain_sum otherwise (LINNE_AMD_FWD_LOSS=0: always the latter); the short Levinson
    trials ride along with the one-unit trial's launch (LINNE_AMD_LEV_RIDE=0: a launch per trial).  Same bytes as the oracle
    every way -- on music, on a signal that makes the last layer choose several units, and with a tail frame the fused
    kernel does not take (so one call runs both forms side by side)"""
    monkeypatch.setenv("LINNE_AMD_FWD_LOSS", fwd_loss)
    monkeypatch.setenv("LINNE_AMD_LEV_RIDE", lev_ride)
    block = 4096
    rng = np.random.default_rng(17)
    t = np.arange(2 * block)
    burst = (6000 * np.sin(2 * np.pi * t * (0.02 + 0.15 * (t // 256 % 2))) * (t // 128 % 2) + rng.integers(-40, 40, size=2 * block)).astype(np.int32)
    x = np.concatenate([music(2, 2 * block, 16, seed=71 + preset), np.stack([burst, burst[::-1] // 3]), music(2, 1234, 16, seed=5)], axis=1)
    mine = product.encode_whole(x, 16, 44100, block, preset, True)
    assert mine == oracle.encode_whole(x, 16, 44100, block, preset, True)
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)


@pytest.mark.parametrize("rows16", ["0", "1"])
@pytest.mark.parametrize("preset", [3, 7])
def test_short_layer_autocorrelation_forms_agree(product, oracle, monkeypatch, rows16, preset):
    """the lags of the order-8 / order-16 layers come from the register-ring form (autocorr_rows) when every unit length of
    the frame is a multiple of 4, from the shared-tile form with LINNE_AMD_ROWS16=0, and from the per-lane form for the
    ragged tail here: same bytes as the oracle either way (LINNE_AMD_L0_PRODUCTS=0 keeps the small batch off the product
    kernel so that these forms run at all)"""
    monkeypatch.setenv("LINNE_AMD_ROWS16", rows16)
    monkeypatch.setenv("LINNE_AMD_L0_PRODUCTS", "0")
    x = music(2, 6 * 4096 + 777, 16, seed=83 + preset)
    mine = product.encode_whole(x, 16, 44100, 4096, preset, True)
    assert mine == oracle.encode_whole(x, 16, 44100, 4096, preset, True)


@pytest.mark.parametrize("hist", ["0", "1"])
@pytest.mark.parametrize("preset,block", [(3, 4096), (7, 4096), (7, 2048)])
def test_long_layer_autocorrelation_forms_agree(product, oracle, monkeypatch, hist, preset, block):
    """the lags of the order-64 / order-128 layer come from the lanes = jobs kernels (k_autocorr_hist, k_autocorr_sub) for the
    frames whose units are whole 16-sample tiles, from k_autocorr2 otherwise (ragged tail; 2048-sample blocks, whose finest
    unit is too short; LINNE_AMD_HIST=0: always) -- one call here runs both side by side: same bytes as the oracle"""
    monkeypatch.setenv("LINNE_AMD_HIST", hist)
    x = music(2, 5 * block + 1001, 16, seed=91 + preset)
    mine = product.encode_whole(x, 16, 44100, block, preset, True)
    assert mine == oracle.encode_whole(x, 16, 44100, block, preset, True)


@pytest.mark.parametrize("hist", ["0", "1"])
def test_batch_with_more_length_runs_than_the_run_table_holds(ctx, oracle, monkeypatch, hist):
    """a batch whose frame lengths alternate (12 runs of 3 classes, all of them lengths the lanes = jobs autocorrelation
    kernels would take): the run table falls back to one run, blocks mix classes, and the general kernels must serve every
    row -- also when the large-batch form is forced"""
    monkeypatch.setenv("LINNE_AMD_HIST", hist)
    nch, bits, block, preset = 2, 16, 10240, 7
    lens = [10240, 8192, 6144] * 4
    frames = np.zeros((len(lens), nch, block), dtype=np.int32)
    for f, n in enumerate(lens):
        frames[f, :, :n] = music(nch, n, bits, seed=300 + f)
    ns = np.array(lens, dtype=np.uint32)
    shape = ctx.shape(nch, bits, block, preset, True)
    res, prm, st = ctx.encode_frames_host(shape, frames, ns)
    for f, n in enumerate(lens):
        enc = oracle.encoder(nch, bits, 44100, block, preset, True)
        tap, ores = enc.hotpath(frames[f][:, :n])
        enc.close()
        _check_taps(tap, prm[f], st[f], preset, nch, f"frame {f} n={n}")
        assert np.array_equal(ores, res[f][:, :n]), f"frame {f}: residual"
    dec = ctx.decode_frames_host(shape, res, prm, ns)
    for f, n in enumerate(lens):
        assert np.array_equal(dec[f][:, :n], frames[f][:, :n])


def test_decode_whole_of_a_stream_with_many_distinct_block_lengths(product, reference):
    """a valid .lnn written by EncodeBlock calls of varying num_samples (linne_encoder.c:774-862 accepts any length up to the
    block size): 48 distinct block lengths in one DecodeWhole group.  The decoder takes the lengths from the stream
    (linne_decoder.c:671-742) and must not care how many different ones there are"""
    import ctypes as C
    from refs import _planar_ptrs, _RefHeader
    nch, bits, block, preset = 2, 16, 4096, 7
    lens = [4096 - 8 * i - (i % 3) for i in range(48)]
    total = sum(lens)
    x = music(nch, total, bits, seed=123)
    enc = reference.new_encoder(nch, bits, 44100, block, preset, True)
    hdr = _RefHeader(1, 2, nch, total, 44100, bits, block, preset, 1)
    out = np.zeros(total * nch * 8 + 65536, dtype=np.uint8)
    assert reference.L.LINNEEncoder_EncodeHeader(C.byref(hdr), out.ctypes.data, out.size) == 0
    off, prog = 30, 0
    for n in lens:
        ptrs, keep = _planar_ptrs(x[:, prog:prog + n])
        osz = C.c_uint32(0)
        assert reference.L.LINNEEncoder_EncodeBlock(enc, ptrs, n, out.ctypes.data + off, out.size - off, C.byref(osz)) == 0
        off += osz.value
        prog += n
    reference.L.LINNEEncoder_Destroy(enc)
    stream = out[:off].tobytes()
    ret, want = reference.decode_whole(stream)
    assert ret == 0 and np.array_equal(want, x)
    ret, got = product.decode_whole(stream)
    assert ret == 0 and np.array_equal(got, x)


def test_corrupt_streams_decode_to_the_reference_pcm(product):
    """CRC check off: wherever the reference decoder accepts a damaged stream (returns OK), this decoder must return OK too
    and deliver the same PCM -- whatever unit counts, shifts, coefficients and residuals the damage produced
    (linne_decoder.c:430-526, linne_lpc_synthesize.c:8-83 with wrap-around int32 arithmetic).  The reference's verdicts
    come from tests/golden/corrupt_decode.json (tests/golden/make_corrupt_golden.py ran the reference decoder built with
    ASan + UBSan in child processes; streams it only survives through undefined behaviour are skipped); streams it rejects
    must be rejected here with the same result code."""
    import hashlib
    import json
    import os
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_corrupt_golden", os.path.join(sys_path, "make_corrupt_golden.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    gold = json.load(open(os.path.join(sys_path, "corrupt_decode.json")))
    x, rng = gen.stream_and_damage()
    good = product.encode_whole(x, 16, 44100, 2048, 7, True)
    assert hashlib.sha256(good).hexdigest() == gold["good_sha256"]
    compared = 0
    for trial, want in enumerate(gold["trials"]):
        bad = gen.damaged(good, rng)
        ret, got = product.decode_whole(bad, check_crc=0)
        assert ret in range(8)
        if not want["defined"]:
            continue                        # the reference only survives this stream through undefined behaviour
        if want["ret"] == 0:
            assert ret == 0, f"trial {trial}: the reference decodes this stream, the product returns {ret}"
            assert gen.fnv_planes(got) == want["fnv"], f"trial {trial}: PCM of the damaged stream differs from the reference decoder's"
            compared += 1
        else:
            assert ret == want["ret"], f"trial {trial}: the reference rejects this stream with {want['ret']}, the product returns {ret}"
    assert compared >= 40


@pytest.mark.parametrize("cap", [None, "64"])
@pytest.mark.parametrize("nch,bits,block,preset,tail", [(2, 16, 10240, 7, 9280), (2, 16, 2048, 4, 777), (1, 24, 4096, 0, 4096), (3, 8, 1024, 2, 1000), (8, 24, 2048, 7, 130)])
def test_device_rice_emission_equals_the_host_coder(ctx, monkeypatch, nch, bits, block, preset, tail, cap):
    """Rice EMISSION on the device (k_rice_scan / k_rice_emit; linne_coder.c:281-302 with bit_stream.h's bit order): the codes the
    device writes, stitched into blocks by LINNEAmd_PackFramesEmitted, must give the bytes of the host coder on the same
    residual -- music, white noise (long codes), tiny residuals (parameter 0), a silent frame, a ragged tail.  With
    LINNE_AMD_RICE_EMIT_CAP=64 nothing fits: every channel-frame takes the fetch-the-residual fallback, same bytes again."""
    import torch
    if cap:
        monkeypatch.setenv("LINNE_AMD_RICE_EMIT_CAP", cap)
    ms = nch >= 2
    F = 7
    frames = music_frames(F, nch, block, bits, seed=131 + nch)
    frames[2] = waveform("white_noise", nch, block, bits, seed=4)
    frames[3] //= 64
    frames[4] = 0
    ns = np.full(F, block, dtype=np.uint32); ns[-1] = tail
    frames[-1, :, tail:] = 0
    shape = ctx.shape(nch, bits, block, preset, ms)
    pcm = torch.from_numpy(frames).cuda()
    res, prm, st = ctx.encode_frames(shape, pcm, ns)
    plan = ctx.rice_plan(shape, res, ns)
    packed, offsets = ctx.rice_emit(shape, res, plan)
    ctx.synchronize()
    res, prm, st, plan, packed, offsets = (t.cpu().numpy() for t in (res, prm, st, plan, packed, offsets))
    want, _ = linne_amd.pack_frames(shape, frames, res, prm, st, ns, 0.0, 2)
    planes = np.ascontiguousarray(frames.transpose(1, 0, 2).reshape(nch, F * block))
    got, _, fetched = linne_amd.pack_frames_emitted(shape, planes, 0, prm, st, plan, packed, offsets, ns, 0.0, 2, residual=res)
    assert got == want
    off = offsets.view(np.uint32)
    if cap:
        assert (off[:-1] == 0xFFFFFFFF).all() and off[-1] == 0 and len(fetched) >= F - 2      # (RAW / SILENT frames need no residual)
    else:
        assert (off[:-1] != 0xFFFFFFFF).all() and not fetched
        nbits = plan[:, :, 4:8].copy().view(np.uint32).reshape(-1)
        assert int(off[-1]) == int(((nbits.astype(np.uint64) + 63) // 64 * 8).sum())


@pytest.mark.parametrize("emit", ["0", "1"])
def test_whole_stream_with_and_without_device_emission(product, oracle, reference, monkeypatch, emit):
    """LINNEEncoder_EncodeWhole stages 16-bit PCM as int16 and lets the device write the Rice codes (LINNE_AMD_EMIT, default 1);
    LINNE_AMD_EMIT=0 keeps int32 staging both ways and the host coder.  Same .lnn either way: the reference's"""
    monkeypatch.setenv("LINNE_AMD_EMIT", emit)
    monkeypatch.setenv("LINNE_AMD_GROUP", "3")
    block = 4096
    parts = [music(2, 7 * block, 16, seed=41), np.zeros((2, block), dtype=np.int32), waveform("white_noise", 2, 2 * block, 16, seed=6),
             music(2, 3 * block + 1500, 16, seed=42)]
    x = np.concatenate(parts, axis=1)
    mine = product.encode_whole(x, 16, 44100, block, 7, True)
    assert mine == reference.encode_whole(x, 16, 44100, block, 7, True)
    x24 = music(3, 5 * 2048 + 99, 24, seed=43)
    assert product.encode_whole(x24, 24, 96000, 2048, 5, True) == oracle.encode_whole(x24, 24, 96000, 2048, 5, True)


@pytest.mark.parametrize("nch,bits,block,preset,total", [(2, 16, 1023, 7, 5 * 1023 + 400), (1, 16, 2047, 4, 3 * 2047 + 1000), (2, 24, 1025, 5, 4 * 1025 + 1021),
                                                         (2, 16, 4095, 7, 2 * 4095 + 4090), (2, 16, 1023, 0, 4000), (8, 16, 1021, 7, 3500), (2, 16, 10239, 7, 3 * 10239 + 5000)])
def test_odd_block_sizes(product, oracle, reference, nch, bits, block, preset, total):
    """an odd num_samples_per_block (linne_encoder.c:410-477 accepts any): the analysis length of a full block is odd, every layer
    has the one-unit trial only, and the Welch window's unwritten middle sample (quirk Q1, lpc.c:200-204) holds what the
    block-type estimate left there -- the last channel's SIN-windowed sample (linne_encoder.c:494-503).  Bytes must equal the
    reference's, whole-stream and block by block; tails of such streams have even analysis lengths and odd sub-lengths"""
    x = music(nch, total, bits, seed=block)
    ms = nch >= 2
    want = reference.encode_whole(x, bits, 44100, block, preset, ms)
    assert oracle.encode_whole(x, bits, 44100, block, preset, ms) == want
    mine = product.encode_whole(x, bits, 44100, block, preset, ms)
    assert mine == want
    assert product.encode_blocks(x, bits, 44100, block, preset, ms) == want
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)


@pytest.mark.parametrize("nch,bits,block,preset,total,af", [(2, 16, 1024, 7, 4 * 1024 + 300, 1), (2, 16, 1024, 7, 4 * 1024 + 300, 3), (1, 16, 2048, 4, 3 * 2048, 2),
                                                            (2, 24, 1024, 0, 3000, 2), (2, 16, 10240, 7, 2 * 10240 + 2000, 1), (2, 16, 1024, 7, 4096, 10),
                                                            (3, 16, 4096, 6, 2 * 4096 + 1001, 2)])
def test_auxiliary_function_iterations(product, oracle, reference, nch, bits, block, preset, total, af):
    """`-a N` (num_afmethod_iterations): the IRLS refinement of every layer's coefficients in the final
    pass (lpc.c:452-509, 578-633; linne_network.c:350-376, 605-630) -- residual reciprocals, the normal matrix as 8384 ordered
    chains, Cholesky with the host's pow(), the final pass run for real.  Bytes equal the reference's"""
    x = music(nch, total, bits, seed=block + af)
    ms = nch >= 2
    want = reference.encode_whole(x, bits, 44100, block, preset, ms, af_iters=af)
    assert oracle.encode_whole(x, bits, 44100, block, preset, ms, af_iters=af) == want
    mine = product.encode_whole(x, bits, 44100, block, preset, ms, af_iters=af)
    assert mine == want
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)


@pytest.mark.parametrize("kind", ["silence", "positive_const", "nyquist", "white_noise", "sine", "chirp"])
def test_auxiliary_function_on_degenerate_signals(product, reference, kind):
    """zero problems (lag 0 below FLT_EPSILON), singular normal matrices, exact fits: the branches of lpc.c:594-618"""
    x = waveform(kind, 2, 4096, 16, seed=1)
    assert product.encode_whole(x, 16, 44100, 1024, 7, True, af_iters=2) == reference.encode_whole(x, 16, 44100, 1024, 7, True, af_iters=2)


@pytest.mark.parametrize("nch,bits,block,preset,total,af", [(1, 16, 1024, 7, 1024 + 300, 0), (2, 16, 1024, 4, 2 * 1024, 0), (2, 16, 512, 0, 1200, 1),
                                                            (1, 24, 1024, 5, 2048, 0), (2, 16, 2048, 7, 2 * 2048 + 999, 2)])
def test_network_trainer(product, oracle, reference, nch, bits, block, preset, total, af):
    """`-l` (enable_learning): LINNENetworkTrainer_Train (linne_network.c:805-873) after the analysis --
    forward, L1 loss, back-propagation through the cascade, momentum step, until the loss stands still -- alone and behind `-a N`.
    Bytes equal the reference's"""
    x = music(nch, total, bits, seed=block + af)
    ms = nch >= 2
    want = reference.encode_whole(x, bits, 44100, block, preset, ms, af_iters=af, learning=1)
    assert oracle.encode_whole(x, bits, 44100, block, preset, ms, af_iters=af, learning=1) == want
    mine = product.encode_whole(x, bits, 44100, block, preset, ms, af_iters=af, learning=1)
    assert mine == want
    ret, dec = product.decode_whole(mine)
    assert ret == 0 and np.array_equal(dec, x)


@pytest.mark.parametrize("kind", ["silence", "nyquist", "sine", "white_noise"])
def test_network_trainer_on_degenerate_signals(product, reference, kind):
    x = waveform(kind, 1, 1024, 16, seed=1)
    assert product.encode_whole(x, 16, 44100, 512, 4, False, learning=1) == reference.encode_whole(x, 16, 44100, 512, 4, False, learning=1)


def _blocks(stream):
    off = 30
    while off + 11 <= len(stream):
        size = int.from_bytes(stream[off + 2:off + 6], "big") + 6
        yield off, size, stream[off + 8]
        off += size


def test_decode_whole_picks_the_rice_decoder_by_the_streams_length(product, monkeypatch):
    """Unless LINNE_AMD_DECODE_STREAM says which, a short stream's Rice codes are decoded by the host threads (the device's decoder is
    serial per block: a latency of 0.29 us per sample of a block, however few blocks) and a long stream's by the device; the PCM is
    the encoder's input either way.  Sixteen host threads, as on the GPU box: 100 stereo blocks of 1024 samples -- the host; 4000 --
    the device; 600 blocks of 4096 samples -- the host, in three groups (decoding overlaps the GPU's work from 1024 channel-frames on)"""
    monkeypatch.delenv("LINNE_AMD_DECODE_STREAM", raising=False)
    monkeypatch.delenv("LINNE_AMD_GROUP", raising=False)
    monkeypatch.setenv("LINNE_AMD_THREADS", "16")
    piece = music(2, 200 * 1024, 16, seed=91)
    for block, frames, tail, want in [(1024, 100, 17, 0), (1024, 4000, 300, 1), (4096, 600, 1000, 0)]:
        total = frames * block + tail
        x = np.ascontiguousarray(np.tile(piece, (1, total // piece.shape[1] + 1))[:, :total])
        stream = product.encode_whole(x, 16, 44100, block, 5, True)
        ret, dec = product.decode_whole(stream)
        assert ret == 0 and np.array_equal(dec, x), f"{frames} blocks of {block}"
        assert (product.last_decode_whole_mode() & 1) == want, f"{frames} blocks of {block}: mode {product.last_decode_whole_mode()}"


@pytest.mark.parametrize("nch,bits,block,preset,total,group", [(2, 16, 4096, 7, 11 * 4096 + 1500, "3"), (1, 16, 2048, 4, 9 * 2048 + 777, "2"), (3, 24, 2048, 5, 6 * 2048 + 99, "4"),
                                                               (8, 8, 1024, 2, 5 * 1024 + 1000, "2"), (2, 16, 1023, 7, 5 * 1023 + 400, "5"), (2, 16, 10240, 7, 40 * 10240 + 9280, None)])
def test_decode_whole_with_rice_decoding_on_the_device(product, monkeypatch, nch, bits, block, preset, total, group):
    """LINNEDecoder_DecodeWhole hands the blocks' bytes to the device, whose k_rice_decode reads the partitioned recursive Rice codes
    (linne_coder.c:304-345) -- LINNE_AMD_DECODE_STREAM=1, the default -- or decodes them on the host threads (0).  Same PCM either
    way (the encoder's input), SILENT and RAW blocks in between, a ragged tail, int16 (<= 16 bits) and int32 PCM on the way back"""
    if group:
        monkeypatch.setenv("LINNE_AMD_GROUP", group)
    parts = [music(nch, total - 3 * block, bits, seed=61 + nch), np.zeros((nch, block), dtype=np.int32), waveform("white_noise", nch, block, bits, seed=8)]
    parts.append(music(nch, total - sum(p.shape[1] for p in parts), bits, seed=62))
    x = np.concatenate(parts, axis=1)
    stream = product.encode_whole(x, bits, 44100, block, preset, nch >= 2)
    for mode in ("1", "0"):
        monkeypatch.setenv("LINNE_AMD_DECODE_STREAM", mode)
        ret, dec = product.decode_whole(stream)
        assert ret == 0 and np.array_equal(dec, x)
        assert product.last_decode_whole_mode() == int(mode)
    # the int32 way back (taken when a sample leaves the 16-bit range: only a stream no encoder wrote decodes to one)
    monkeypatch.setenv("LINNE_AMD_DECODE_STREAM", "1")
    monkeypatch.setenv("LINNE_AMD_DEBUG_NO_PCM16", "1")
    ret, dec = product.decode_whole(stream)
    assert ret == 0 and np.array_equal(dec, x) and product.last_decode_whole_mode() == 1
    monkeypatch.delenv("LINNE_AMD_DEBUG_NO_PCM16")
    # a stream cut inside a block: the error is reported, the blocks before it are delivered
    ret, dec = product.decode_whole(stream[:len(stream) * 2 // 3])
    whole = 0
    for off, size, _ in _blocks(stream):
        if off + size > len(stream) * 2 // 3:
            break
        whole += 1
    assert ret == 4 and np.array_equal(dec[:, :whole * block], x[:, :whole * block])


def test_decode_whole_meets_blocks_no_encoder_writes(product, oracle, monkeypatch):
    """Blocks whose CRC is right but whose payload is not an encoder's (here: damaged, then the CRC recomputed): the device's Rice
    decoder either reads them as the host's does or reports them (orders above 10, codes running past the block, a block that
    ends before its size field says), and the call starts over on the host -- the result is the host decoder's in every case"""
    import ctypes as C
    monkeypatch.setenv("LINNE_AMD_GROUP", "3")
    block = 2048
    x = music(2, 12 * block + 500, 16, seed=77)
    good = product.encode_whole(x, 16, 44100, block, 7, True)
    rng = np.random.default_rng(5)
    restarted = same = 0
    for trial in range(24):
        bad = bytearray(good)
        blocks = [b for b in _blocks(good) if b[2] == 0]
        for off, size, _ in [blocks[i] for i in rng.choice(len(blocks), size=2, replace=False)]:
            lo = off + 11 + (0 if trial % 3 == 0 else size // 3)                # now and then inside the parameters
            for pos in rng.integers(lo, off + size, size=int(rng.integers(1, 4))):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            body = np.frombuffer(bytes(bad[off + 8:off + size]), dtype=np.uint8)
            bad[off + 6:off + 8] = int(oracle.L.oracle_crc16(body.ctypes.data, len(body))).to_bytes(2, "big")
        bad = bytes(bad)
        monkeypatch.setenv("LINNE_AMD_DECODE_STREAM", "0")
        want = product.decode_whole(bad)
        monkeypatch.setenv("LINNE_AMD_DECODE_STREAM", "1")
        got = product.decode_whole(bad)
        mode = product.last_decode_whole_mode()
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), f"trial {trial}"
        restarted += (mode & 2) != 0
        same += (mode == 1)
    assert restarted > 0 and same > 0


@pytest.mark.parametrize("wide,lev_wave", [("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")])
@pytest.mark.parametrize("nch,bits,block,preset,tail", [(2, 16, 10240, 7, 9280), (1, 24, 4096, 3, 1000), (2, 16, 1024, 0, 512), (2, 8, 2048, 5, 777)])
def test_block_at_a_time_forms_agree(product, oracle, monkeypatch, wide, lev_wave, nch, bits, block, preset, tail):
    """LINNEEncoder_EncodeBlock, block by block as the reference's tools/linne_codec calls it (linne_codec.c:133-161): a handful of
    jobs per call, served by the latency forms -- k_autocorr_prod / k_autocorr_wide (LINNE_AMD_WIDE), k_levinson_wave
    (LINNE_AMD_LEV_WAVE), k_chain_sum_wave -- or, with the knobs off, by the batch forms on the same few jobs.  Every block's
    bytes equal the oracle's encoder's, the ragged last block included"""
    import ctypes as C
    monkeypatch.setenv("LINNE_AMD_WIDE", wide)
    monkeypatch.setenv("LINNE_AMD_LEV_WAVE", lev_wave)
    ms = nch >= 2
    x = music(nch, 3 * block + tail, bits, seed=300 + preset)
    want = oracle.encode_whole(x, bits, 44100, block, preset, ms)
    enc = product.new_encoder(nch, bits, 44100, block, preset, ms)
    out = np.zeros(nch * block * 8 + 65536, dtype=np.uint8)
    osz = C.c_uint32(0)
    got = bytearray(want[:30])                                      # (the header carries the total length: EncodeWhole's business)
    pos = 0
    while pos < x.shape[1]:
        n = min(block, x.shape[1] - pos)
        planes = [np.ascontiguousarray(x[ch, pos:pos + n]) for ch in range(nch)]
        ptrs = (C.POINTER(C.c_int32) * nch)(*[pl.ctypes.data_as(C.POINTER(C.c_int32)) for pl in planes])
        assert product.L.LINNEEncoder_EncodeBlock(enc, ptrs, n, out.ctypes.data, out.size, C.byref(osz)) == 0
        got += bytes(out[:osz.value])
        pos += n
    product.L.LINNEEncoder_Destroy(enc)
    assert bytes(got) == want


@pytest.mark.gpu
def test_the_references_full_round_trip_matrix(product):
    """every cell of the reference's own round-trip matrix (test/linne_encode_decode/main.cpp:335-536: 9 waveforms x {1,2,8}
    channels x {8,16,24} bits x presets {0,7}, 8192 samples in blocks of 1024): the bytes hash to what oracle/_ref wrote
    (tests/golden/matrix_hashes.json, committed: no _ref needed here) and DecodeWhole restores the input, as the reference's test
    asks"""
    import hashlib
    from test_oracle_cpu import matrix_cells
    n = 0
    for name, x, bits, preset, ms, e in matrix_cells():
        got = product.encode_whole(x, bits, 8000, 1024, preset, ms)
        assert len(got) == e["bytes"] and hashlib.sha256(got).hexdigest() == e["sha256"], name
        ret, dec = product.decode_whole(got)
        assert ret == 0 and np.array_equal(dec, x), name
        n += 1
    assert n == 162


@pytest.mark.gpu
def test_option_goldens_without_the_reference_library(product):
    """-a 1/2/3 and -l (lpc.c:578-633, linne_network.c:805-873) pinned by committed hashes of oracle/_ref's streams: these stay
    tests (not skips) on a box where oracle/_ref was not shipped"""
    import hashlib
    from test_oracle_cpu import option_cases
    n = 0
    for name, x, e in option_cases():
        bits = e["music_args"][2]
        got = product.encode_whole(x, bits, 44100, e["block"], e["preset"], x.shape[0] >= 2, af_iters=e["af_iters"], learning=e["learning"])
        assert len(got) == e["bytes"] and hashlib.sha256(got).hexdigest() == e["sha256"], name
        ret, dec = product.decode_whole(got)
        assert ret == 0 and np.array_equal(dec, x), name
        n += 1
    assert n == 9


@pytest.mark.gpu
def test_two_handles_on_two_threads_at_once(product, oracle):
    """the boundary's threading contract (a handle is used by one thread at a time, different handles are independent):
    two encoder handles and two decoder handles driven from two threads at the same moment, whole streams and block-at-a-time
    calls mixed, several rounds -- every result equals the single-threaded one"""
    import threading
    cases = [(music(2, 9 * 4096 + 1234, 16, seed=401), 16, 4096, 7, True), (music(1, 7 * 2048 + 99, 24, seed=402), 24, 2048, 4, False)]
    want = [oracle.encode_whole(x, bits, 44100, block, preset, ms) for x, bits, block, preset, ms in cases]
    errors = []
    gate = threading.Barrier(2)

    def work(k):
        x, bits, block, preset, ms = cases[k]
        try:
            for rnd in range(4):
                gate.wait(timeout=120)
                got = product.encode_whole(x, bits, 44100, block, preset, ms) if rnd % 2 == 0 else product.encode_blocks(x, bits, 44100, block, preset, ms)
                assert got == want[k], f"thread {k} round {rnd}: encode differs"
                ret, dec = product.decode_whole(got)
                assert ret == 0 and np.array_equal(dec, x), f"thread {k} round {rnd}: decode differs"
        except Exception as exc:          # (a failed assert on one side must not leave the other at the barrier)
            errors.append(repr(exc))
            gate.abort()

    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not errors, errors


@pytest.mark.gpu
def test_rice_decode_device_reads_no_further_than_its_contract(ctx):
    """LINNEAmd_RiceDecodeDevice promises to read d_stream only up to the next multiple of 8 bytes behind stream_bytes
    (include/linne_amd.h): a mono batch's emitted codes in a buffer that ends exactly there -- with non-zero bytes between
    stream_bytes and that bound -- decode to the residual, and every frame's end position is the next frame's start"""
    import ctypes as C
    import torch
    nch, bits, block, preset, F = 1, 16, 4096, 4, 5
    frames = music_frames(F, nch, block, bits, seed=77)
    frames[2] = waveform("white_noise", nch, block, bits, seed=3)
    ns = np.full(F, block, dtype=np.uint32); ns[-1] = 3000
    frames[-1, :, 3000:] = 0
    shape = ctx.shape(nch, bits, block, preset, False)
    res, prm, st = ctx.encode_frames(shape, torch.from_numpy(frames).cuda(), ns)
    plan = ctx.rice_plan(shape, res, ns)
    packed, offsets = ctx.rice_emit(shape, res, plan)
    ctx.synchronize()
    off = offsets.cpu().numpy().view(np.uint32)
    assert (off != 0xFFFFFFFF).all()
    nbits = plan.cpu().numpy()[:, 0, linne_amd.RICE_PLAN_NBITS:linne_amd.RICE_PLAN_NBITS + 4].copy().view(np.uint32)[:, 0]
    for k in (F, F - 1, F - 2, F - 3):                       # prefixes of the batch: stream_bytes on and off the 4- and 8-byte grids
        total = int(off[k])
        bound = (total + 7) & ~7
        buf = torch.full((bound,), 0xA5, dtype=torch.uint8, device="cuda")
        buf[:total] = packed[:total]
        bitpos = torch.from_numpy((off[:k].astype(np.uint64) * 8).view(np.int64)).cuda()
        out = torch.full((k, nch, block), 123456, dtype=torch.int32, device="cuda")
        endbit = torch.zeros(k, dtype=torch.int64, device="cuda")
        nsk = np.ascontiguousarray(ns[:k])
        ctx._fence()         # (the context has a stream of its own: torch's fills and copies above must be done first)
        ret = linne_amd.lib.LINNEAmd_RiceDecodeDevice(C.c_void_p(ctx.h), C.byref(shape), C.c_void_p(buf.data_ptr()), C.c_uint64(total), C.c_void_p(bitpos.data_ptr()),
                                                      nsk.ctypes.data_as(C.c_void_p), C.c_uint32(k), C.c_void_p(out.data_ptr()), C.c_void_p(endbit.data_ptr()))
        assert ret == 0
        ctx.synchronize()
        got, want, eb = out.cpu().numpy(), res.cpu().numpy(), endbit.cpu().numpy().view(np.uint64)
        for f in range(k):
            assert np.array_equal(got[f, 0, :ns[f]], want[f, 0, :ns[f]]), f"{k} frames, frame {f}"
            assert int(eb[f]) == int(off[f]) * 8 + int(nbits[f]), f"{k} frames, frame {f}: end position"


@pytest.mark.parametrize("stats_rows,prep_general,prep_defer", [("1", "0", "1"), ("1", "0", "0"), ("0", "1", "1"), ("1", "1", "0")])
def test_prep_and_statistics_forms_agree(ctx, product, oracle, monkeypatch, stats_rows, prep_general, prep_defer):
    """k_prep keeps the channel in registers (blocks up to 10 240 samples, exact-integer correlations) or streams it through
    global memory (LINNE_AMD_PREP_GENERAL=1; any length; loud 24-bit material takes the ordered double chains either way: in
    k_prep_slow, lanes = channel-frames, or with LINNE_AMD_PREP_DEFER=0 on two lanes of k_prep's own block); the
    block-type statistics come from k_stats (a block per channel-frame) or k_stats_rows (lanes = channel-frames;
    LINNE_AMD_STATS_ROWS).  Whatever the form: the oracle's pre-emphasis, SIN-window r0, parameters and residual on full
    frames, ragged tails, both layer-0 orders (presets 0 / 7), 3 and 8 channels, loud 24-bit, and the oracle's bytes across
    SILENT / RAW / COMPRESS blocks (the decision reads the statistics)"""
    monkeypatch.setenv("LINNE_AMD_STATS_ROWS", stats_rows)
    monkeypatch.setenv("LINNE_AMD_PREP_GENERAL", prep_general)
    monkeypatch.setenv("LINNE_AMD_PREP_DEFER", prep_defer)
    for nch, bits, block, preset, tail, loud in [(2, 16, 10240, 7, 9280, False), (1, 16, 1024, 0, 130, False), (8, 24, 4096, 7, 1000, True),
                                                 (3, 8, 1024, 2, 1023, False), (2, 16, 2048, 1, 2048, False), (2, 24, 10240, 5, 681, True)]:
        ms = nch >= 2
        F = 5
        frames = music_frames(F, nch, block, bits, seed=500 + block + preset)
        if loud:
            frames = np.clip(frames.astype(np.int64) * 3, -(1 << (bits - 1)), (1 << (bits - 1)) - 1).astype(np.int32)
        frames[2] = waveform("chirp", nch, block, bits, seed=9)
        ns = np.full(F, block, dtype=np.uint32); ns[-1] = tail
        frames[-1, :, tail:] = 0
        shape = ctx.shape(nch, bits, block, preset, ms)
        res, prm, st = ctx.encode_frames_host(shape, frames, ns)
        enc = oracle.encoder(nch, bits, 44100, block, preset, ms)
        for f in range(F):
            n = int(ns[f])
            tap, ores = enc.hotpath(frames[f][:, :n])
            _check_taps(tap, prm[f], st[f], preset, nch, f"{nch}ch {bits}bit block {block} -m {preset} frame {f}")
            assert np.array_equal(ores, res[f][:, :n])
        enc.close()
    block = 2048
    x = np.concatenate([music(2, 5 * block, 16, seed=31), np.zeros((2, 2 * block), dtype=np.int32), waveform("white_noise", 2, 2 * block, 16, seed=5),
                        music(2, 3 * block + 555, 16, seed=32)], axis=1)
    for preset in (7, 0):
        assert product.encode_whole(x, 16, 44100, block, preset, True) == oracle.encode_whole(x, 16, 44100, block, preset, True)
    # 24-bit material travels to the GPU as packed 3-byte samples: both statistics kernels (and k_prep) unpack it themselves
    x24 = np.concatenate([music(2, 4 * block + 300, 24, seed=44), np.zeros((2, block), dtype=np.int32), music(2, 2 * block + 17, 24, seed=45) * 5], axis=1)
    x24 = np.clip(x24, -(1 << 23), (1 << 23) - 1).astype(np.int32)
    for preset in (7, 1):
        assert product.encode_whole(x24, 24, 96000, block, preset, True) == oracle.encode_whole(x24, 24, 96000, block, preset, True)


@pytest.mark.parametrize("prep_defer", ["1", "0"])
def test_loud_and_quiet_frames_in_one_call(ctx, oracle, monkeypatch, prep_defer):
    """k_prep_slow's list: 71 stereo frames of 24-bit material in one call, every other one loud -- the pre-emphasis sums of both its
    channels leave the exact-integer range (ordered chains); of the quiet ones only the side channel's do: about a hundred listed rows
    between rows k_prep finishes itself = two blocks of the kernel, the second partly filled; frames of ten lengths in one block of rows
    (ends inside a tile, on a tile's edge, one sample off it), one of them three samples long, one a single sample (no product at
    all); silence among the loud ones.  The oracle's parameters (pre-emphasis
    coefficients and first samples among them) and residual, frame by frame"""
    monkeypatch.setenv("LINNE_AMD_PREP_DEFER", prep_defer)
    nch, bits, block, preset, F = 2, 24, 4096, 5, 71
    frames = music_frames(F, nch, block, bits, seed=77)
    loud = np.zeros(F, dtype=bool); loud[::2] = True
    frames[loud] = np.clip(frames[loud].astype(np.int64) * 6, -(1 << (bits - 1)), (1 << (bits - 1)) - 1).astype(np.int32)
    frames[4] = 0
    ns = np.full(F, block, dtype=np.uint32); ns[10] = 3; ns[12] = 1; ns[14] = 1001; ns[16] = 2048; ns[18] = 64; ns[20] = 65; ns[22] = 63; ns[24] = 4095; ns[-1] = 2047
    for f in range(F): frames[f, :, ns[f]:] = 0
    shape = ctx.shape(nch, bits, block, preset, True)
    res, prm, st = ctx.encode_frames_host(shape, frames, ns)
    enc = oracle.encoder(nch, bits, 44100, block, preset, True)
    for f in range(F):
        n = int(ns[f])
        tap, ores = enc.hotpath(frames[f][:, :n])
        _check_taps(tap, prm[f], st[f], preset, nch, f"frame {f} ({'loud' if loud[f] else 'quiet'}, {n} samples)")
        assert np.array_equal(ores, res[f][:, :n]), f"frame {f}"
    enc.close()


def test_host_libm_values_on_the_gpu_box():
    """SURVEY 7.5 item 6 on the box the parity tests run on: its libm, and the library's host code, give the committed Welch divisors,
    SIN-window samples and Cholesky pivots bit for bit (tests/golden/libm_values.json) -- a mismatch here explains 162 differing hashes"""
    from test_abi_cpu import test_host_libm_values_are_the_build_containers
    test_host_libm_values_are_the_build_containers()


def test_decode_refuses_a_coefficient_outside_the_formats_range(ctx):
    """The stream's coefficients are 8-bit Huffman symbols (libs/linne_decoder/src/linne_decoder.c:452-470): [-128, 127].  The
    throughput / latency forms of the synthesis carry them as int8, the lanes form would take any int32 -- so a parameter record with a
    coefficient outside the range must be REFUSED where the host can see it (LINNEAmd_DecodeFramesHost), whatever the batch size
    would have picked; the untouched batch still decodes."""
    nch, bits, block, preset = 2, 16, 2048, 7
    frames = music_frames(3, nch, block, bits, seed=2718)
    shape = ctx.shape(nch, bits, block, preset, True)
    res, prm, st = ctx.encode_frames_host(shape, frames)
    assert np.array_equal(ctx.decode_frames_host(shape, res, prm), frames)
    for kernel in (None, "lanes", "rows"):
        for value in (128, -129, 70000):
            bad = prm.copy()
            bad[1, 1, linne_amd.PARAM_WORDS - 20] = value          # a coefficient of the last layer, second frame, second channel
            old = os.environ.pop("LINNE_AMD_DECODE_KERNEL", None)
            if kernel:
                os.environ["LINNE_AMD_DECODE_KERNEL"] = kernel
            try:
                with pytest.raises(linne_amd.LinneAmdError, match="-> 2"):
                    ctx.decode_frames_host(shape, res, bad)
            finally:
                os.environ.pop("LINNE_AMD_DECODE_KERNEL", None)
                if old is not None:
                    os.environ["LINNE_AMD_DECODE_KERNEL"] = old
