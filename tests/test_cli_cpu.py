"""CPU tests (-m "not gpu") of the command line tool (tools/cli, built to linne_amd/linne_amd_cli): option handling as
the reference's tools/linne_codec/linne_codec.c:285-395 does it, and the WAV writer on streams whose blocks are all
RAW / SILENT (those decode on the host alone)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from refs import ROOT

CLI = os.path.join(ROOT, "linne_amd", "linne_amd_cli")
pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="linne_amd_cli not built (python -c 'import __graft_entry__ as g; g.build()')")


def run(*args):
    return subprocess.run([CLI, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def wav_bytes(x, bits, rate):
    """canonical 44-byte-header PCM WAV of planar int samples x [C][N] (what libs/wav/src/wav.c:523-760 writes)"""
    nch, n = x.shape
    nb = bits // 8
    inter = x.T.reshape(-1)
    if bits == 8:
        data = (inter + 128).astype(np.uint8).tobytes()
    elif bits == 16:
        data = inter.astype("<i2").tobytes()
    elif bits == 24:
        u = inter.astype(np.int64) & 0xFFFFFF
        data = np.stack([u & 255, (u >> 8) & 255, (u >> 16) & 255], axis=1).astype(np.uint8).tobytes()
    else:
        data = inter.astype("<i4").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, nch, rate, rate * nb * nch, nb * nch, bits)
    return hdr + b"data" + struct.pack("<I", len(data)) + data


def test_usage_version_and_option_errors():
    r = run()
    assert r.returncode == 1 and "Usage:" in r.stdout
    assert run("-h").returncode == 0 and "--mode" in run("-h").stdout
    assert "Version.2" in run("-v").stdout
    assert run("-e", "-d", "a", "b").returncode == 1
    assert run("a", "b").returncode == 1                               # neither -e nor -d
    assert run("-e", "a").returncode == 1                              # one file name only
    assert run("-e", "-m", "8", "a", "b").returncode == 1              # preset out of range
    assert run("-e", "--mode=9", "a", "b").returncode == 1
    assert run("-e", "-x", "a", "b").returncode == 1                   # unknown option
    r = run("-d", "/nonexistent/in.lnn", "/tmp/out.wav")
    assert r.returncode == 1 and "Failed to open" in r.stderr


def test_refinement_options_are_accepted(tmp_path):
    """-l and -a N are served by the device path (lnn_k_train.h, lnn_k_af.h): SetEncodeParameter accepts them, and without a GPU
    the encode then fails loudly like any other (no CPU fallback)"""
    x = np.zeros((1, 4096), dtype=np.int32)
    w = tmp_path / "z.wav"
    w.write_bytes(wav_bytes(x, 16, 8000))
    for extra in (["-l"], ["-a", "2"], ["--auxiliary-function-iteration=1"]):
        r = run("-e", "-m", "3", *extra, str(w), str(tmp_path / "z.lnn"))
        assert "Failed to set encode parameter" not in r.stderr
        assert r.returncode == 0 or "no CPU fallback" in r.stderr


@pytest.mark.parametrize("nch,bits", [(2, 16), (1, 8), (3, 24)])
def test_decode_host_only_streams_to_wav(oracle, tmp_path, nch, bits):
    """SILENT and RAW blocks decode without the GPU: .lnn (from the oracle) -> WAV must be the canonical file"""
    rng = np.random.default_rng(nch * 10 + bits)
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    x = np.concatenate([np.zeros((nch, 1500), dtype=np.int32), rng.integers(lo, hi, size=(nch, 2500)).astype(np.int32)], axis=1)
    # block size 500: the first three blocks are silent, the rest noise (raw)
    lnn = oracle.encode_whole(x, bits, 22050, 500, 4, nch >= 2)
    types, off = [], 30
    while off < len(lnn):
        types.append(lnn[off + 8]); off += int.from_bytes(lnn[off + 2:off + 6], "big") + 6
    assert set(types) <= {1, 2} and 1 in types and 2 in types
    src, dst = tmp_path / "a.lnn", tmp_path / "a.wav"
    src.write_bytes(lnn)
    r = run("-d", str(src), str(dst))
    assert r.returncode == 0, r.stderr
    assert dst.read_bytes() == wav_bytes(x, bits, 22050)
    # batch form: names keep their stem
    out = tmp_path / "out"; out.mkdir()
    r = run("-d", "-q", "--batch", str(out), str(src))
    assert r.returncode == 0 and (out / "a.wav").read_bytes() == wav_bytes(x, bits, 22050)
    # a corrupted payload is reported (CRC16), and -c skips the check
    bad = bytearray(lnn); bad[-1] ^= 0x55
    src.write_bytes(bytes(bad))
    assert run("-d", str(src), str(dst)).returncode == 1
    assert run("-d", "-c", str(src), str(dst)).returncode == 0
