"""Generates the golden vectors of tests/golden/ with the REAL reference (oracle/_ref/liblinne_ref.so, built by
oracle/Makefile from /root/reference).  Run in the build container:  python tests/golden/make_golden.py

Outputs (data only -- inputs and expected outputs):
  golden_streams.npz   per case: input int32 [ch][n], stream parameters, the reference's .lnn bytes
  matrix_hashes.json   sha256 of the reference's .lnn for every cell of the reference's own 162-case round-trip matrix
                       (test/linne_encode_decode/main.cpp:335-536) and for nine small streams encoded with -a N / -l
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


# the reference's full round-trip matrix (test/linne_encode_decode/main.cpp:335-536): 9 waveforms x {1, 2, 8 ch} x {8, 16, 24 bit} x
# preset {0, 7}, 8192 samples, block 1024, rate 8000; MS for more than one channel except for silence, whose rows all say NONE
MATRIX_KINDS = ["silence", "sine", "antiphase_sine", "white_noise", "chirp", "positive_const", "negative_const", "nyquist", "gauss_noise"]


def matrix_cases():
    for kind in MATRIX_KINDS:
        for preset in (0, 7):
            for nch in (1, 2, 8):
                for bits in (8, 16, 24):
                    yield kind, nch, bits, preset, int(nch >= 2 and kind != "silence")


# -a N and -l (lpc.c:578-633, linne_network.c:805-873): two small streams each, so that the -m gpu suite pins them without oracle/_ref
OPT_CASES = [  # name, (nch, ns, bits, seed), block, preset, af_iters, learning
    ("a1_2ch16_m7", (2, 2 * 1024 + 300, 16, 41), 1024, 7, 1, 0), ("a1_1ch24_m4", (1, 2048 + 77, 24, 42), 2048, 4, 1, 0),
    ("a2_2ch16_m7", (2, 2 * 1024 + 300, 16, 43), 1024, 7, 2, 0), ("a2_3ch16_m2", (3, 1024 + 512, 16, 44), 512, 2, 2, 0),
    ("a3_2ch16_m7", (2, 2 * 1024 + 300, 16, 45), 1024, 7, 3, 0), ("a3_1ch8_m0", (1, 4096 + 100, 8, 46), 2048, 0, 3, 0),
    ("l_2ch16_m4", (2, 2 * 1024, 16, 47), 1024, 4, 0, 1), ("l_1ch16_m7", (1, 1024 + 300, 16, 48), 1024, 7, 0, 1),
    ("a1l_2ch16_m0", (2, 1200, 16, 49), 512, 0, 1, 1),
]


def main():
    ref = Reference()
    matrix = {}
    for kind, nch, bits, preset, ms in matrix_cases():
        x = waveform(kind, nch, 8192, bits, seed=nch * 100 + bits)
        lnn = ref.encode_whole(x, bits, 8000, 1024, preset, bool(ms))
        ret, dec = ref.decode_whole(lnn)
        assert ret == 0 and np.array_equal(dec, x)          # what the reference's own test checks
        matrix[f"{kind}/{nch}ch/{bits}bit/m{preset}"] = {"bytes": len(lnn), "sha256": hashlib.sha256(lnn).hexdigest(), "ms": ms,
                                                         "input_sha256": hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()}
    opt = {}
    for name, margs, block, preset, af, learn in OPT_CASES:
        x = music(*margs)
        lnn = ref.encode_whole(x, margs[2], 44100, block, preset, margs[0] >= 2, af_iters=af, learning=learn)
        opt[name] = {"bytes": len(lnn), "sha256": hashlib.sha256(lnn).hexdigest(), "input_sha256": hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest(), "music_args": list(margs), "block": block, "preset": preset,
                     "af_iters": af, "learning": learn}
    json.dump({"matrix": matrix, "options": opt}, open(os.path.join(HERE, "matrix_hashes.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(matrix), "matrix cells,", len(opt), "-a / -l streams")
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
