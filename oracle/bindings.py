"""ctypes bindings of the two checkers (TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg):

* ``Oracle``   -- oracle/liblinne_oracle.so, this repo's CPU restatement (oracle/linne_oracle.c)
* ``Reference`` -- oracle/_ref/liblinne_ref.so, the real reference compiled by oracle/Makefile from
  /root/reference (present only where that build ran; the .so travels to the GPU box, the sources do not),
  bound through the same public-API binding the product uses (linne_amd.api.LinneApi)

Nothing in linne_amd/ imports this module.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE_SO = os.path.join(ROOT, "oracle", "liblinne_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "liblinne_ref.so")

MAX_CH, MAX_LAYERS, MAX_PARAMS = 8, 3, 128
PRESET_LAYERS = {0: (2, 32), 1: (2, 32), 2: (4, 64, 8), 3: (4, 64, 8), 4: (4, 64, 8),
                 5: (4, 128, 16), 6: (4, 128, 16), 7: (4, 128, 16)}


from linne_amd.api import _planar_ptrs  # noqa: E402


class EncodeParameter(C.Structure):
    _fields_ = [("num_channels", C.c_uint32), ("bits_per_sample", C.c_uint32), ("sampling_rate", C.c_uint32),
                ("num_samples_per_block", C.c_uint32), ("preset", C.c_uint32), ("ch_process_method", C.c_uint32)]


class ChannelTap(C.Structure):
    _fields_ = [("preem_prev", C.c_int32 * 2), ("preem_coef", C.c_int32 * 2),
                ("num_units", C.c_uint32 * MAX_LAYERS), ("rshift", C.c_uint32 * MAX_LAYERS),
                ("coef", (C.c_int32 * MAX_PARAMS) * MAX_LAYERS),
                ("coef_double", (C.c_double * MAX_PARAMS) * MAX_LAYERS),
                ("pass_loss", C.c_double * 4), ("best_pass", C.c_uint32),
                ("est_r0", C.c_double), ("est_parcor", C.c_double * (MAX_PARAMS + 2)),
                ("est_length", C.c_double), ("parcor_tail", C.c_double)]


class FrameTap(C.Structure):
    _fields_ = [("block_type", C.c_uint32), ("num_samples", C.c_uint32), ("num_analyze_samples", C.c_uint32),
                ("ch", ChannelTap * MAX_CH)]


def oracle_available():
    return os.path.exists(ORACLE_SO)


def reference_available():
    return os.path.exists(REF_SO)


class Oracle:
    def __init__(self):
        L = C.CDLL(ORACLE_SO)
        L.oracle_encoder_create.restype = C.c_void_p
        L.oracle_encoder_create.argtypes = [C.POINTER(EncodeParameter)]
        L.oracle_encoder_destroy.argtypes = [C.c_void_p]
        L.oracle_encoder_set_af_iterations.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_encoder_set_learning.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                          C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        L.oracle_encode_whole.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.oracle_encode_frame_hotpath.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_decode_whole.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
        L.oracle_decode_frame_hotpath.argtypes = [C.POINTER(EncodeParameter), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.oracle_crc16.restype = C.c_uint16
        L.oracle_crc16.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_rice_encode.restype = C.c_uint32
        L.oracle_rice_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.oracle_rice_decode.restype = C.c_uint32
        L.oracle_rice_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.oracle_huffman_code.restype = C.c_uint32
        L.oracle_huffman_code.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
        L.oracle_bench_encode.restype = C.c_double
        L.oracle_bench_encode.argtypes = [C.POINTER(EncodeParameter), C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        self.L = L

    @staticmethod
    def param(nch, bits, rate, block, preset, ms):
        return EncodeParameter(nch, bits, rate, block, preset, int(ms))

    def encode_whole(self, x, bits, rate, block, preset, ms, af_iters=0, learning=0):
        x = np.ascontiguousarray(x, dtype=np.int32)
        p = self.param(x.shape[0], bits, rate, block, preset, ms)
        enc = self.L.oracle_encoder_create(C.byref(p))
        assert enc, "oracle_encoder_create failed"
        self.L.oracle_encoder_set_af_iterations(enc, af_iters)
        self.L.oracle_encoder_set_learning(enc, learning)
        ptrs, keep = _planar_ptrs(x)
        cap = x.size * 4 * 2 + 65536
        out = np.zeros(cap, dtype=np.uint8)
        osz = C.c_uint32(0)
        ret = self.L.oracle_encode_whole(enc, ptrs, x.shape[1], out.ctypes.data, cap, C.byref(osz))
        self.L.oracle_encoder_destroy(enc)
        assert ret == 0, f"oracle_encode_whole -> {ret}"
        return out[:osz.value].tobytes()

    def decode_whole(self, data, check_crc=1):
        buf = np.frombuffer(data, dtype=np.uint8)
        hdr = np.zeros(9, dtype=np.uint32)
        nch = int.from_bytes(data[12:14], "big")
        ns = int.from_bytes(data[14:18], "big")
        out = np.zeros((nch, ns), dtype=np.int32)
        ret = self.L.oracle_decode_whole(buf.ctypes.data, len(data), out.ctypes.data, nch, ns, hdr.ctypes.data, check_crc)
        return ret, out, hdr

    class Encoder:
        """stateful block encoder (keeps the reference's cross-block buffer state, quirks Q1/Q2)"""

        def __init__(self, oracle, nch, bits, rate, block, preset, ms, af_iters=0):
            self.o = oracle
            self.p = Oracle.param(nch, bits, rate, block, preset, ms)
            self.h = oracle.L.oracle_encoder_create(C.byref(self.p))
            assert self.h
            oracle.L.oracle_encoder_set_af_iterations(self.h, af_iters)
            self.nch, self.block = nch, block

        def close(self):
            if self.h:
                self.o.L.oracle_encoder_destroy(self.h)
                self.h = None

        def __del__(self):
            self.close()

        def encode_block(self, x):
            """x [ch][n] -> (bytes, FrameTap, residual[ch][n] or None)"""
            x = np.ascontiguousarray(x, dtype=np.int32)
            n = x.shape[1]
            ptrs, keep = _planar_ptrs(x)
            cap = x.size * 4 * 2 + 65536
            out = np.zeros(cap, dtype=np.uint8)
            osz = C.c_uint32(0)
            tap = FrameTap()
            res = np.zeros((self.nch, self.block), dtype=np.int32)
            ret = self.o.L.oracle_encode_block(self.h, ptrs, n, out.ctypes.data, cap, C.byref(osz), C.byref(tap), res.ctypes.data)
            assert ret == 0, ret
            return out[:osz.value].tobytes(), tap, res[:, :n]

        def hotpath(self, x):
            x = np.ascontiguousarray(x, dtype=np.int32)
            n = x.shape[1]
            tap = FrameTap()
            res = np.zeros_like(x)
            ret = self.o.L.oracle_encode_frame_hotpath(self.h, x.ctypes.data, n, n, C.byref(tap), res.ctypes.data)
            assert ret == 0, ret
            return tap, res

    def encoder(self, nch, bits, rate, block, preset, ms, af_iters=0):
        return Oracle.Encoder(self, nch, bits, rate, block, preset, ms, af_iters)

    def decode_hotpath(self, taps, residual, bits, block, preset, ms):
        """taps: sequence of ChannelTap (one per channel); residual [ch][n] -> pcm [ch][n]"""
        d = np.ascontiguousarray(residual, dtype=np.int32).copy()
        nch, n = d.shape
        p = self.param(nch, bits, 44100, block, preset, ms)
        arr = (ChannelTap * nch)(*taps)
        ret = self.L.oracle_decode_frame_hotpath(C.byref(p), arr, d.ctypes.data, n, n)
        assert ret == 0
        return d



def Reference():
    from linne_amd.api import LinneApi
    return LinneApi(REF_SO)


def fnv1a64(b):
    h = 0xcbf29ce484222325
    for byte in b:
        h ^= byte
        h = (h * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h
