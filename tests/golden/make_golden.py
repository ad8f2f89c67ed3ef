"""Generates the golden vectors of tests/golden/ with the REAL reference (oracle/_ref/liblinne_ref.so, built by
oracle/Makefile from /root/reference).  Run in the build container:  python tests/golden/make_golden.py

Outputs (data only -- inputs and expected outputs):
  golden_streams.npz   per case: input int32 [ch][n], stream parameters, the reference's .lnn bytes
  golden_hashes.json   sha256 of the reference's .lnn for the larger cases (inputs regenerated from signals.py or
                       read from the two WAV data files copied from the reference's own test fixtures)
  ref_a.wav, ref_16bit_2ch.wav   data files of the reference's tests (test/linne_internal/a.wav with its CRC16
                       known answer 0xA611, test/wav/16bit_2ch.wav), used as real-audio inputs
"""
import hashlib
import json
import os
import sys
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from refs import Reference  # noqa: E402
from signals import music, waveform  # noqa: E402


def read_wav(path):
    w = wave.open(path, "rb")
    nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
    raw = np.frombuffer(w.readframes(n), dtype=np.uint8)
    if width == 2:
        x = raw.view("<i2").reshape(n, nch).T.astype(np.int32)
    else:
        assert width == 1                       # 8-bit WAV is offset binary
        x = raw.reshape(n, nch).T.astype(np.int32) - 128
    return np.ascontiguousarray(x), rate, width * 8


SMALL = [  # kind, nch, bits, preset, ms   (8192 samples, block 1024, rate 8000: the reference's round-trip matrix shape)
    ("sine", 1, 16, 0, 0), ("sine", 2, 16, 7, 1), ("chirp", 2, 16, 7, 1), ("chirp", 2, 24, 7, 1), ("antiphase_sine", 8, 16, 7, 1),
    ("sine", 1, 8, 4, 0), ("chirp", 1, 16, 4, 0), ("white_noise", 2, 16, 7, 1), ("silence", 2, 16, 7, 1), ("nyquist", 1, 8, 0, 0),
    ("gauss_noise", 2, 16, 4, 1), ("negative_const", 2, 24, 5, 1),
]
LARGE = [  # name, generator args: music(nch, ns, bits, seed), block, preset, ms, rate
    ("music_2ch16_m7_tail2000", (2, 2 * 10240 + 2000, 16, 21), 10240, 7, 1, 44100),
    ("music_2ch16_m7_tail680", (2, 10240 + 680, 16, 22), 10240, 7, 1, 44100),
    ("music_2ch16_m7_tail9280", (2, 10240 + 9280, 16, 23), 10240, 7, 1, 44100),
    ("music_1ch16_m4_tail680", (1, 3 * 10240 + 680, 16, 24), 10240, 4, 0, 44100),
    ("music_8ch24_m7", (8, 10240 + 2000, 24, 25), 10240, 7, 1, 96000),
    ("music_2ch16_m0_block4096", (2, 3 * 4096 + 1001, 16, 26), 4096, 0, 1, 44100),
]


def main():
    ref = Reference()
    out, hashes = {}, {}
    for i, (kind, nch, bits, preset, ms) in enumerate(SMALL):
        x = waveform(kind, nch, 8192, bits, seed=nch * 100 + bits)
        lnn = ref.encode_whole(x, bits, 8000, 1024, preset, bool(ms))
        out[f"s{i}_x"] = x.astype(np.int32)
        out[f"s{i}_meta"] = np.array([bits, 8000, 1024, preset, ms], dtype=np.int64)
        out[f"s{i}_lnn"] = np.frombuffer(lnn, dtype=np.uint8)
        hashes[f"small/{kind}_{nch}ch_{bits}b_m{preset}"] = {"bytes": len(lnn), "sha256": hashlib.sha256(lnn).hexdigest()}
    for name, margs, block, preset, ms, rate in LARGE:
        x = music(*margs)
        lnn = ref.encode_whole(x, margs[2], rate, block, preset, bool(ms))
        hashes[f"large/{name}"] = {"bytes": len(lnn), "sha256": hashlib.sha256(lnn).hexdigest(),
                                    "input_sha256": hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest(),
                                    "music_args": list(margs), "block": block, "preset": preset, "ms": ms, "rate": rate}
    for fn, preset in (("ref_a.wav", 7), ("ref_16bit_2ch.wav", 7), ("ref_16bit_2ch.wav", 4)):
        x, rate, bits = read_wav(os.path.join(HERE, fn))
        ms = x.shape[0] >= 2
        lnn = ref.encode_whole(x, bits, rate, 10240, preset, ms)
        hashes[f"wav/{fn}_m{preset}"] = {"bytes": len(lnn), "sha256": hashlib.sha256(lnn).hexdigest(), "channels": int(x.shape[0]),
                                          "samples": int(x.shape[1]), "rate": rate, "bits": bits, "block": 10240, "preset": preset, "ms": int(ms)}
    np.savez_compressed(os.path.join(HERE, "golden_streams.npz"), **out)
    json.dump(hashes, open(os.path.join(HERE, "golden_hashes.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(SMALL), "small streams,", len(hashes), "hashes")


if __name__ == "__main__":
    main()
