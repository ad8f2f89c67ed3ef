"""Deterministic test / benchmark inputs.

* the waveform families the reference's round-trip matrix uses (test/linne_encode_decode/main.cpp:48-189:
  silence, sine, anti-phase sine, white noise, chirp, +/- full scale constant, Nyquist, Gaussian noise),
  restated with numpy and quantised as main.cpp:192-214 does;
* the compressible synthetic "music" of SURVEY.md section 8(d): harmonics + AR(2) noise, so that the encoder
  takes the COMPRESS path (white noise would be emitted as RAW blocks and skip the hot path).
"""
import numpy as np


def _quantize(x, bits):
    """main.cpp:192-214: round(x * 2^(bits-1)) with the positive side clipped to 2^(bits-1)-1"""
    v = np.asarray(x, dtype=np.float64) * float(1 << (bits - 1))
    q = np.where(v >= 0.0, np.floor(v + 0.5), -np.floor(-v + 0.5)).astype(np.int64)   # LINNEUtility_Round
    return np.minimum(q, (1 << (bits - 1)) - 1).astype(np.int32)


def waveform(kind, nch, ns, bits, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(ns, dtype=np.float64)
    if kind == "silence":
        x = np.zeros((nch, ns))
    elif kind == "sine":
        x = np.tile(np.sin(880.0 * np.pi * t / 44100.0), (nch, 1))            # main.cpp:72
    elif kind == "antiphase_sine":
        x = np.stack([((-1.0) ** ch) * np.sin(880.0 * np.pi * t / 44100.0) for ch in range(nch)])   # main.cpp:87
    elif kind == "white_noise":
        x = rng.uniform(-1.0, 1.0, size=(nch, ns))
    elif kind == "chirp":
        x = np.tile(np.sin((2.0 * np.pi * t) / (ns - t)), (nch, 1))            # main.cpp:118-119
    elif kind == "positive_const":
        x = np.ones((nch, ns))
    elif kind == "negative_const":
        x = -np.ones((nch, ns))
    elif kind == "nyquist":
        x = np.tile(np.where((np.arange(ns) & 1) == 0, 1.0, -1.0), (nch, 1))
    elif kind == "gauss_noise":
        x = np.clip(rng.normal(0.0, 0.25, size=(nch, ns)), -1.0, 1.0)         # main.cpp:184 (sigma 0.25)
    else:
        raise ValueError(kind)
    return _quantize(x, bits)


WAVEFORMS = ["silence", "sine", "antiphase_sine", "white_noise", "chirp", "positive_const",
             "negative_const", "nyquist", "gauss_noise"]


def music(nch, ns, bits, seed=0, rate=44100.0):
    """6 harmonics of f0 = 110*(ch+1) Hz, amplitudes 0.3/(k+1), random phases, plus AR(2) noise
    (poles a1=1.6, a2=-0.8, sigma 0.05, normalised to 0.3 peak-ish), mix 0.6*tone + noise, clip, round."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(0x4C494E4E ^ seed)
    t = np.arange(ns, dtype=np.float64) / rate
    out = np.zeros((nch, ns))
    for ch in range(nch):
        f0 = 110.0 * (ch + 1)
        tone = np.zeros(ns)
        for k in range(6):
            tone += (0.3 / (k + 1)) * np.sin(2 * np.pi * f0 * (k + 1) * t + rng.uniform(0, 2 * np.pi))
        e = rng.normal(0.0, 0.05, size=ns)
        noise = lfilter([1.0], [1.0, -1.6, 0.8], e)
        noise *= 0.3 / max(1e-9, np.max(np.abs(noise)))
        out[ch] = np.clip(0.6 * tone + noise, -0.999, 0.999)
    return _quantize(out, bits)


def music_frames(num_frames, nch, block, bits, seed=0):
    """[frame][ch][block] int32 of independent 'music' frames (each frame its own seed)"""
    x = music(nch, num_frames * block, bits, seed)
    return np.ascontiguousarray(x.reshape(nch, num_frames, block).transpose(1, 0, 2))
