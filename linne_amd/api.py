"""linne_amd.api -- ctypes binding of the 13-function LINNE public C API (include/linne_encoder.h, include/linne_decoder.h)
as exported by a shared library that implements it: liblinne_amd.so (this package's product, the default) or -- in tests and in
bench.py's cpu_baseline leg -- the real reference compiled under oracle/_ref.  Host buffers in, host buffers out; nothing here
computes anything."""
import ctypes as C

import numpy as np


def _planar_ptrs(x):
    """x: int32 array [ch][n] (C-contiguous rows) -> (int32_t*[ch], keepalive)"""
    x = np.ascontiguousarray(x, dtype=np.int32)
    ptrs = (C.POINTER(C.c_int32) * x.shape[0])()
    for ch in range(x.shape[0]):
        ptrs[ch] = x[ch].ctypes.data_as(C.POINTER(C.c_int32))
    return ptrs, x



class _RefHeader(C.Structure):
    _fields_ = [("format_version", C.c_uint32), ("codec_version", C.c_uint32), ("num_channels", C.c_uint16),
                ("num_samples", C.c_uint32), ("sampling_rate", C.c_uint32), ("bits_per_sample", C.c_uint16),
                ("num_samples_per_block", C.c_uint32), ("preset", C.c_uint8), ("ch_process_method", C.c_int)]


class _RefEncodeParameter(C.Structure):
    _fields_ = [("num_channels", C.c_uint16), ("bits_per_sample", C.c_uint16), ("sampling_rate", C.c_uint32),
                ("num_samples_per_block", C.c_uint16), ("preset", C.c_uint8), ("ch_process_method", C.c_int),
                ("enable_learning", C.c_uint8), ("num_afmethod_iterations", C.c_uint8)]


class _RefEncoderConfig(C.Structure):
    _fields_ = [("max_num_channels", C.c_uint32), ("max_num_samples_per_block", C.c_uint32),
                ("max_num_layers", C.c_uint32), ("max_num_parameters_per_layer", C.c_uint32)]


class _RefDecoderConfig(C.Structure):
    _fields_ = [("max_num_channels", C.c_uint32), ("max_num_layers", C.c_uint32),
                ("max_num_parameters_per_layer", C.c_uint32), ("check_crc", C.c_uint8)]


class LinneApi:
    """Binding of the public LINNE C API (include/linne_encoder.h, include/linne_decoder.h) as exported by
    any shared library that implements it -- the real reference or this repo's drop-in."""

    def __init__(self, so_path):
        L = C.CDLL(so_path)
        L.LINNEEncoder_Create.restype = C.c_void_p
        L.LINNEEncoder_Create.argtypes = [C.POINTER(_RefEncoderConfig), C.c_void_p, C.c_int32]
        L.LINNEEncoder_CalculateWorkSize.restype = C.c_int32
        L.LINNEEncoder_CalculateWorkSize.argtypes = [C.POINTER(_RefEncoderConfig)]
        L.LINNEEncoder_Destroy.argtypes = [C.c_void_p]
        L.LINNEEncoder_SetEncodeParameter.argtypes = [C.c_void_p, C.POINTER(_RefEncodeParameter)]
        L.LINNEEncoder_EncodeHeader.argtypes = [C.POINTER(_RefHeader), C.c_void_p, C.c_uint32]
        L.LINNEEncoder_EncodeBlock.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.LINNEEncoder_EncodeWhole.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.LINNEDecoder_Create.restype = C.c_void_p
        L.LINNEDecoder_Create.argtypes = [C.POINTER(_RefDecoderConfig), C.c_void_p, C.c_int32]
        L.LINNEDecoder_CalculateWorkSize.restype = C.c_int32
        L.LINNEDecoder_CalculateWorkSize.argtypes = [C.POINTER(_RefDecoderConfig)]
        L.LINNEDecoder_Destroy.argtypes = [C.c_void_p]
        L.LINNEDecoder_DecodeHeader.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_RefHeader)]
        L.LINNEDecoder_SetHeader.argtypes = [C.c_void_p, C.POINTER(_RefHeader)]
        L.LINNEDecoder_DecodeBlock.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.LINNEDecoder_DecodeWhole.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
        self.L = L

    Header, EncodeParameter, EncoderConfig, DecoderConfig = _RefHeader, _RefEncodeParameter, _RefEncoderConfig, _RefDecoderConfig

    def new_encoder(self, nch, bits, rate, block, preset, ms, max_block=None, af_iters=0, learning=0):
        cfg = _RefEncoderConfig(max(nch, 1), max_block or block, 5, 128)
        # The reference mallocs its work area and never clears the LPC calculator's buffers; what the first
        # block's raw/compress decision reads there (oracle quirk Q2) is heap garbage unless the caller
        # supplies the memory.  Tests hand it zero-filled memory, which is also what a fresh process gets
        # from a large malloc.
        wsize = self.L.LINNEEncoder_CalculateWorkSize(C.byref(cfg))
        assert wsize > 0
        work = np.zeros(wsize + 64, dtype=np.uint8)
        enc = self.L.LINNEEncoder_Create(C.byref(cfg), work.ctypes.data, wsize + 64)
        assert enc, "LINNEEncoder_Create failed"
        self._work = getattr(self, "_work", {})
        self._work[enc] = work
        par = _RefEncodeParameter(nch, bits, rate, block, preset, int(ms), learning, af_iters)
        ret = self.L.LINNEEncoder_SetEncodeParameter(enc, C.byref(par))
        if ret != 0:
            self.L.LINNEEncoder_Destroy(enc)
            raise RuntimeError(f"SetEncodeParameter -> {ret}")
        return enc

    def encode_whole(self, x, bits, rate, block, preset, ms, af_iters=0, learning=0):
        x = np.ascontiguousarray(x, dtype=np.int32)
        enc = self.new_encoder(x.shape[0], bits, rate, block, preset, ms, af_iters=af_iters, learning=learning)
        ptrs, keep = _planar_ptrs(x)
        cap = x.size * 4 * 2 + 65536
        out = np.zeros(cap, dtype=np.uint8)
        osz = C.c_uint32(0)
        ret = self.L.LINNEEncoder_EncodeWhole(enc, ptrs, x.shape[1], out.ctypes.data, cap, C.byref(osz))
        self.L.LINNEEncoder_Destroy(enc)
        assert ret == 0, f"EncodeWhole -> {ret}"
        return out[:osz.value].tobytes()

    def encode_blocks(self, x, bits, rate, block, preset, ms):
        """header + one EncodeBlock call per block, as tools/linne_codec/linne_codec.c:123-161 does"""
        x = np.ascontiguousarray(x, dtype=np.int32)
        nch, ns = x.shape
        enc = self.new_encoder(nch, bits, rate, block, preset, ms)
        hdr = _RefHeader(1, 2, nch, ns, rate, bits, block, preset, int(ms))
        cap = x.size * 4 * 2 + 65536
        out = np.zeros(cap, dtype=np.uint8)
        ret = self.L.LINNEEncoder_EncodeHeader(C.byref(hdr), out.ctypes.data, cap)
        assert ret == 0
        off, prog = 30, 0
        while prog < ns:
            n = min(block, ns - prog)
            ptrs, keep = _planar_ptrs(x[:, prog:prog + n])
            osz = C.c_uint32(0)
            ret = self.L.LINNEEncoder_EncodeBlock(enc, ptrs, n, out.ctypes.data + off, cap - off, C.byref(osz))
            assert ret == 0, ret
            off += osz.value
            prog += n
        self.L.LINNEEncoder_Destroy(enc)
        return out[:off].tobytes()

    def decode_whole(self, data, check_crc=1):
        buf = np.frombuffer(data, dtype=np.uint8)
        nch = int.from_bytes(data[12:14], "big")
        ns = int.from_bytes(data[14:18], "big")
        cfg = _RefDecoderConfig(max(nch, 1), 5, 128, check_crc)
        dec = self.L.LINNEDecoder_Create(C.byref(cfg), None, 0)
        assert dec
        out = np.zeros((max(nch, 1), max(ns, 1)), dtype=np.int32)
        ptrs, keep = _planar_ptrs(out)
        ret = self.L.LINNEDecoder_DecodeWhole(dec, buf.ctypes.data, len(data), ptrs, out.shape[0], out.shape[1])
        self.L.LINNEDecoder_Destroy(dec)
        return ret, keep

    def last_decode_whole_mode(self):
        """liblinne_amd only: bit 0 = the last decode_whole let the device decode the Rice codes, bit 1 = it started over on the host"""
        fn = self.L.LINNEAmd_LastDecodeWholeMode
        fn.restype = C.c_uint32
        return int(fn())




def product_api():
    """the public API as exported by liblinne_amd.so"""
    from . import LIB_PATH
    return LinneApi(LIB_PATH)
