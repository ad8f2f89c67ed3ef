"""CPU tests (-m "not gpu") of the product's HOST stage (linne_amd/csrc/lnn_entropy.c): given the hot-path
results the oracle computes (standing in for the GPU), LINNEAmd_PackFrames must emit the oracle's blocks byte for
byte -- block-type decision (incl. quirk Q2), Huffman-coded parameters, partitioned Rice code, block header, CRC."""
import numpy as np
import pytest

import linne_amd
from signals import music, waveform


def taps_to_arrays(tap, res, nch, preset, block):
    prm = np.zeros((nch, linne_amd.PARAM_WORDS), dtype=np.int32)
    st = np.zeros((nch, linne_amd.STAT_WORDS), dtype=np.float64)
    P0 = linne_amd.PRESET_LAYERS[preset][0]
    for ch in range(nch):
        t = tap.ch[ch]
        prm[ch, 0:2] = list(t.preem_prev); prm[ch, 2:4] = list(t.preem_coef)
        nl = len(linne_amd.PRESET_LAYERS[preset])
        prm[ch, 4:4 + nl] = list(t.num_units)[:nl]; prm[ch, 7:7 + nl] = list(t.rshift)[:nl]
        off = 10
        for l, P in enumerate(linne_amd.PRESET_LAYERS[preset]):
            prm[ch, off:off + P] = list(t.coef[l][:P]); off += P
        st[ch, linne_amd.ST_R0] = t.est_r0
        # the estimate's own parcor[1..P0-1]; an all-zero vector with tiny r0 is the zero branch
        pc = list(t.est_parcor)
        st[ch, linne_amd.ST_K1:linne_amd.ST_K1 + 3] = (pc[1:4] + [0, 0, 0])[:3]
        st[ch, linne_amd.ST_ZERO] = 1.0 if abs(t.est_r0) < 1.1920928955078125e-07 else 0.0
        st[ch, linne_amd.ST_TAIL] = t.parcor_tail
    full = np.zeros((nch, block), dtype=np.int32)
    full[:, :res.shape[1]] = res
    return prm, st, full


@pytest.mark.parametrize("nch,bits,block,preset,ms", [(2, 16, 1024, 7, True), (1, 16, 2048, 4, False), (2, 24, 1024, 0, True), (3, 8, 1024, 2, False)])
def test_pack_frames_matches_oracle_blocks(oracle, nch, bits, block, preset, ms):
    rng = np.random.default_rng(5)
    parts = [music(nch, 3 * block, bits, seed=3), np.zeros((nch, block), dtype=np.int32),
             waveform("white_noise", nch, 2 * block, bits, seed=9), music(nch, block + block // 3, bits, seed=4)]
    x = np.concatenate(parts, axis=1)
    ns_total = x.shape[1]
    F = (ns_total + block - 1) // block
    enc = oracle.encoder(nch, bits, 44100, block, preset, ms)
    pcm = np.zeros((F, nch, block), dtype=np.int32)
    prm = np.zeros((F, nch, linne_amd.PARAM_WORDS), dtype=np.int32)
    st = np.zeros((F, nch, linne_amd.STAT_WORDS), dtype=np.float64)
    res = np.zeros((F, nch, block), dtype=np.int32)
    nsm = np.zeros(F, dtype=np.uint32)
    want = []
    for f in range(F):
        seg = x[:, f * block:(f + 1) * block]
        n = seg.shape[1]
        nsm[f] = n
        pcm[f, :, :n] = seg
        blk, tap, r = enc.encode_block(seg)
        want.append(blk)
        if tap.block_type != 0:
            # RAW / SILENT blocks carry no analysis in the oracle's tap: run the hot path on a scratch handle for the
            # statistics only (the real GPU path analyses every frame speculatively)
            e2 = oracle.encoder(nch, bits, 44100, block, preset, ms)
            tap2, r2 = e2.hotpath(seg)
            e2.close()
            for ch in range(nch):
                tap2.ch[ch].est_r0 = tap.ch[ch].est_r0
                for i in range(8):
                    tap2.ch[ch].est_parcor[i] = tap.ch[ch].est_parcor[i]
            tap, r = tap2, r2
        prm[f], st[f], res[f] = taps_to_arrays(tap, r, nch, preset, block)
    enc.close()
    shape = linne_amd.Shape(nch, bits, block, preset, int(ms))
    for threads in (1, 4):
        blocks, state = linne_amd.pack_frames(shape, pcm, res, prm, st, nsm, 0.0, threads)
        types = [b[8] for b in blocks]
        assert types == [w[8] for w in want], f"block types {types}"
        assert 0 in types and 1 in types and 2 in types               # compress, silent and raw all occur
        for f in range(F):
            assert blocks[f] == want[f], f"frame {f} (type {types[f]}) differs"


def test_entropy_stage_under_sanitizers(tmp_path):
    """tools/entropy_fuzz.c: the host bit I/O, Rice coder and block parser under AddressSanitizer + UBSan -- random blocks of
    every type round-trip exactly (incl. full-range residuals, where the parameter reaches 31) and damaged blocks parse
    without an out-of-bounds access.  GPU sanitizers do not exist on the pool; this is the CPU build the task asks for."""
    import os, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "entropy_fuzz")
    cc = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
          "-I" + os.path.join(root, "linne_amd", "csrc"), "-I" + os.path.join(root, "include"),
          os.path.join(root, "tools", "entropy_fuzz.c"), "-lm", "-lpthread", "-o", exe]
    b = subprocess.run(cc, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe, "6", "11"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no sanitizer report" in r.stdout
