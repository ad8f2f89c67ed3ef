"""CPU tests (-m "not gpu") of the drop-in boundary: liblinne_amd.so loads, exports every symbol include/*.h
declares, and keeps the reference's argument / ownership / error conventions (test/linne_encoder/
linne_encoder_test.cpp:49-458, test/linne_decoder/linne_decoder_test.cpp:74-579) for everything that needs no GPU.
No compute call is made here; the one that would need the GPU must fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import linne_amd
from refs import LinneApi, _RefDecoderConfig, _RefEncodeParameter, _RefEncoderConfig, _RefHeader
from signals import music

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OK, INVALID_ARGUMENT, INVALID_FORMAT, INSUFFICIENT_BUFFER, INSUFFICIENT_DATA, PARAMETER_NOT_SET, CORRUPTION, NG = range(8)


@pytest.fixture(scope="module")
def api():
    return LinneApi(linne_amd.LIB_PATH)


def valid_header():
    return _RefHeader(1, 2, 2, 1024, 44100, 16, 1024, 0, 1)


def test_library_exports_every_declared_symbol():
    declared = set()
    for fn in ("linne_encoder.h", "linne_decoder.h", "linne_amd.h"):
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b(LINNE(?:Encoder|Decoder|Amd)_\w+)\s*\(", src))
    assert len(declared) >= 28
    assert declared == set(linne_amd.API_SYMBOLS) | set(linne_amd.AMD_SYMBOLS)
    for name in sorted(declared):
        assert hasattr(linne_amd.lib, name), f"{name} is declared in include/ but not exported"


def test_encode_header_layout_and_errors(api):
    L = api.L
    data = np.zeros(30, dtype=np.uint8)
    h = valid_header()
    assert L.LINNEEncoder_EncodeHeader(C.byref(h), data.ctypes.data, 30) == OK
    assert bytes(data[:4]) == b"IBRA"
    assert list(data[4:12]) == [0, 0, 0, 1, 0, 0, 0, 2] and list(data[12:14]) == [0, 2]
    assert int.from_bytes(bytes(data[14:18]), "big") == 1024 and int.from_bytes(bytes(data[18:22]), "big") == 44100
    assert int.from_bytes(bytes(data[22:24]), "big") == 16 and int.from_bytes(bytes(data[24:28]), "big") == 1024 and list(data[28:30]) == [0, 1]
    assert L.LINNEEncoder_EncodeHeader(None, data.ctypes.data, 30) == INVALID_ARGUMENT
    assert L.LINNEEncoder_EncodeHeader(C.byref(h), None, 30) == INVALID_ARGUMENT
    assert L.LINNEEncoder_EncodeHeader(C.byref(h), data.ctypes.data, 29) == INSUFFICIENT_BUFFER
    for field, bad in [("num_channels", 0), ("num_samples", 0), ("sampling_rate", 0), ("bits_per_sample", 0),
                       ("num_samples_per_block", 0), ("preset", 8), ("ch_process_method", 2)]:
        h = valid_header()
        setattr(h, field, bad)
        assert L.LINNEEncoder_EncodeHeader(C.byref(h), data.ctypes.data, 30) == INVALID_FORMAT, field
    h = valid_header()
    h.num_channels = 1                                  # MS on mono
    assert L.LINNEEncoder_EncodeHeader(C.byref(h), data.ctypes.data, 30) == INVALID_FORMAT
    # the decoder reads back what the encoder wrote
    h = valid_header()
    L.LINNEEncoder_EncodeHeader(C.byref(h), data.ctypes.data, 30)
    g = _RefHeader()
    assert L.LINNEDecoder_DecodeHeader(data.ctypes.data, 30, C.byref(g)) == OK
    assert (g.format_version, g.codec_version, g.num_channels, g.num_samples, g.sampling_rate, g.bits_per_sample,
            g.num_samples_per_block, g.preset, g.ch_process_method) == (1, 2, 2, 1024, 44100, 16, 1024, 0, 1)
    assert L.LINNEDecoder_DecodeHeader(data.ctypes.data, 29, C.byref(g)) == INSUFFICIENT_DATA
    data[0] = ord("X")
    assert L.LINNEDecoder_DecodeHeader(data.ctypes.data, 30, C.byref(g)) == INVALID_FORMAT


def test_encoder_create_destroy_contracts(api):
    L = api.L
    cfg = _RefEncoderConfig(8, 16384, 5, 128)
    size = L.LINNEEncoder_CalculateWorkSize(C.byref(cfg))
    assert size > 0
    assert L.LINNEEncoder_CalculateWorkSize(None) == -1
    for field in ("max_num_channels", "max_num_samples_per_block", "max_num_layers", "max_num_parameters_per_layer"):
        bad = _RefEncoderConfig(8, 16384, 5, 128)
        setattr(bad, field, 0)
        assert L.LINNEEncoder_CalculateWorkSize(C.byref(bad)) == -1
        assert not L.LINNEEncoder_Create(C.byref(bad), None, 0)
    assert L.LINNEEncoder_CalculateWorkSize(C.byref(_RefEncoderConfig(2, 64, 3, 128))) == -1    # params > block
    enc = L.LINNEEncoder_Create(C.byref(cfg), None, 0)                                          # library-owned work area
    assert enc
    L.LINNEEncoder_Destroy(enc)
    work = np.zeros(size + 16, dtype=np.uint8)
    enc = L.LINNEEncoder_Create(C.byref(cfg), work.ctypes.data, size)                           # caller-owned work area
    assert enc and work.ctypes.data <= enc < work.ctypes.data + size
    L.LINNEEncoder_Destroy(enc)
    assert not L.LINNEEncoder_Create(C.byref(cfg), work.ctypes.data, size - 1)
    assert not L.LINNEEncoder_Create(C.byref(cfg), None, size)
    assert not L.LINNEEncoder_Create(None, work.ctypes.data, size)
    L.LINNEEncoder_Destroy(None)


def test_set_encode_parameter_errors(api):
    L = api.L
    cfg = _RefEncoderConfig(2, 4096, 3, 128)
    enc = L.LINNEEncoder_Create(C.byref(cfg), None, 0)
    good = lambda: _RefEncodeParameter(2, 16, 44100, 1024, 7, 1, 0, 0)
    assert L.LINNEEncoder_SetEncodeParameter(None, C.byref(good())) == INVALID_ARGUMENT
    assert L.LINNEEncoder_SetEncodeParameter(enc, None) == INVALID_ARGUMENT
    assert L.LINNEEncoder_SetEncodeParameter(enc, C.byref(good())) == OK
    for field, bad, want in [("num_channels", 0, INVALID_FORMAT), ("bits_per_sample", 0, INVALID_FORMAT), ("sampling_rate", 0, INVALID_FORMAT),
                             ("num_samples_per_block", 0, INVALID_FORMAT), ("preset", 8, INVALID_FORMAT), ("ch_process_method", 2, INVALID_FORMAT),
                             ("num_samples_per_block", 100, INVALID_FORMAT),       # block <= largest layer (128 at -m 7)
                             ("num_channels", 3, INSUFFICIENT_BUFFER), ("num_samples_per_block", 8192, INSUFFICIENT_BUFFER),
                             ("enable_learning", 1, OK),                                # -l is served (lnn_k_train.h)
                             ("num_afmethod_iterations", 2, OK)]:                       # -a N is served (lnn_k_af.h)
        prm = good()
        setattr(prm, field, bad)
        assert L.LINNEEncoder_SetEncodeParameter(enc, C.byref(prm)) == want, field
    L.LINNEEncoder_Destroy(enc)
    small = L.LINNEEncoder_Create(C.byref(_RefEncoderConfig(2, 4096, 2, 32)), None, 0)      # preset 7 needs 3 layers / 128 params
    assert L.LINNEEncoder_SetEncodeParameter(small, C.byref(good())) == INSUFFICIENT_BUFFER
    L.LINNEEncoder_Destroy(small)


def test_encode_block_argument_errors_and_loud_failure_without_gpu(api, capfd):
    L = api.L
    enc = L.LINNEEncoder_Create(C.byref(_RefEncoderConfig(2, 4096, 3, 128)), None, 0)
    x = music(2, 1024, 16, seed=1)
    ptrs = (C.POINTER(C.c_int32) * 2)(*[x[ch].ctypes.data_as(C.POINTER(C.c_int32)) for ch in range(2)])
    out = np.zeros(65536, dtype=np.uint8)
    osz = C.c_uint32(0)
    assert L.LINNEEncoder_EncodeBlock(enc, ptrs, 1024, out.ctypes.data, out.size, C.byref(osz)) == PARAMETER_NOT_SET
    assert L.LINNEEncoder_EncodeWhole(enc, ptrs, 1024, out.ctypes.data, out.size, C.byref(osz)) == PARAMETER_NOT_SET
    prm = _RefEncodeParameter(2, 16, 44100, 1024, 7, 1, 0, 0)
    assert L.LINNEEncoder_SetEncodeParameter(enc, C.byref(prm)) == OK
    assert L.LINNEEncoder_EncodeBlock(None, ptrs, 1024, out.ctypes.data, out.size, C.byref(osz)) == INVALID_ARGUMENT
    assert L.LINNEEncoder_EncodeBlock(enc, None, 1024, out.ctypes.data, out.size, C.byref(osz)) == INVALID_ARGUMENT
    assert L.LINNEEncoder_EncodeBlock(enc, ptrs, 0, out.ctypes.data, out.size, C.byref(osz)) == INVALID_ARGUMENT
    assert L.LINNEEncoder_EncodeBlock(enc, ptrs, 1024, None, out.size, C.byref(osz)) == INVALID_ARGUMENT
    assert L.LINNEEncoder_EncodeBlock(enc, ptrs, 1024, out.ctypes.data, 0, C.byref(osz)) == INVALID_ARGUMENT
    assert L.LINNEEncoder_EncodeBlock(enc, ptrs, 1024, out.ctypes.data, out.size, None) == INVALID_ARGUMENT
    assert L.LINNEEncoder_EncodeBlock(enc, ptrs, 1025, out.ctypes.data, out.size, C.byref(osz)) == INSUFFICIENT_BUFFER
    if linne_amd.device_count() == 0:
        # no GPU: the prediction path must not silently run elsewhere
        assert L.LINNEEncoder_EncodeBlock(enc, ptrs, 1024, out.ctypes.data, out.size, C.byref(osz)) == NG
        assert "no CPU fallback" in capfd.readouterr().err
        with pytest.raises(linne_amd.LinneAmdError):
            linne_amd.Context(0, use_torch_stream=False)
    L.LINNEEncoder_Destroy(enc)


def test_decoder_contracts_and_host_only_blocks(api, oracle):
    L = api.L
    cfg = _RefDecoderConfig(8, 5, 128, 1)
    size = L.LINNEDecoder_CalculateWorkSize(C.byref(cfg))
    assert size > 0 and L.LINNEDecoder_CalculateWorkSize(None) == -1
    for field in ("max_num_channels", "max_num_layers", "max_num_parameters_per_layer"):
        bad = _RefDecoderConfig(8, 5, 128, 1)
        setattr(bad, field, 0)
        assert L.LINNEDecoder_CalculateWorkSize(C.byref(bad)) == -1 and not L.LINNEDecoder_Create(C.byref(bad), None, 0)
    work = np.zeros(size + 16, dtype=np.uint8)
    assert not L.LINNEDecoder_Create(C.byref(cfg), work.ctypes.data, size - 1)
    dec = L.LINNEDecoder_Create(C.byref(cfg), work.ctypes.data, size)
    assert dec
    buf = np.zeros((2, 2048), dtype=np.int32)
    ptrs = (C.POINTER(C.c_int32) * 2)(*[buf[ch].ctypes.data_as(C.POINTER(C.c_int32)) for ch in range(2)])
    dsz, dn = C.c_uint32(0), C.c_uint32(0)
    blk = np.zeros(64, dtype=np.uint8)
    assert L.LINNEDecoder_DecodeBlock(dec, blk.ctypes.data, 64, ptrs, 2, 2048, C.byref(dsz), C.byref(dn)) == PARAMETER_NOT_SET
    h = valid_header()
    for field, bad in [("format_version", 2), ("codec_version", 3), ("num_channels", 0), ("preset", 8), ("ch_process_method", 2)]:
        g = valid_header()
        setattr(g, field, bad)
        assert L.LINNEDecoder_SetHeader(dec, C.byref(g)) == INVALID_FORMAT, field
    assert L.LINNEDecoder_SetHeader(None, C.byref(h)) == INVALID_ARGUMENT
    assert L.LINNEDecoder_SetHeader(dec, C.byref(h)) == OK
    # a SILENT and a RAW stream decode without touching the GPU (linne_decoder_test.cpp:395-468)
    for x in (np.zeros((2, 2048), dtype=np.int32), np.random.default_rng(0).integers(-32768, 32767, size=(2, 2048)).astype(np.int32)):
        lnn = oracle.encode_whole(x, 16, 44100, 1024, 7, True)
        assert lnn[30 + 8] in (1, 2)                                   # block type: silent / raw
        ret, out = api.decode_whole(lnn)
        assert ret == OK and np.array_equal(out, x)
        data = np.frombuffer(lnn, dtype=np.uint8)
        assert L.LINNEDecoder_DecodeBlock(dec, data[30:].ctypes.data, len(lnn) - 30, ptrs, 2, 2048, C.byref(dsz), C.byref(dn)) == OK
        assert dn.value == 1024 and np.array_equal(buf[:, :1024], x[:, :1024])
        assert L.LINNEDecoder_DecodeBlock(dec, data[30:].ctypes.data, len(lnn) - 30, ptrs, 1, 2048, C.byref(dsz), C.byref(dn)) == INSUFFICIENT_BUFFER
        assert L.LINNEDecoder_DecodeBlock(dec, data[30:].ctypes.data, len(lnn) - 30, ptrs, 2, 1000, C.byref(dsz), C.byref(dn)) == INSUFFICIENT_BUFFER
        bad = data[30:].copy(); bad[0] ^= 0xFF
        assert L.LINNEDecoder_DecodeBlock(dec, bad.ctypes.data, bad.size, ptrs, 2, 2048, C.byref(dsz), C.byref(dn)) == INVALID_FORMAT
        if lnn[30 + 8] == 2:
            bad = data[30:].copy(); bad[20] ^= 0x55
            assert L.LINNEDecoder_DecodeBlock(dec, bad.ctypes.data, bad.size, ptrs, 2, 2048, C.byref(dsz), C.byref(dn)) == CORRUPTION
            assert L.LINNEDecoder_DecodeBlock(dec, data[30:].ctypes.data, 100, ptrs, 2, 2048, C.byref(dsz), C.byref(dn)) == INSUFFICIENT_DATA
    L.LINNEDecoder_Destroy(dec)


def libm_values_of_this_box():
    """(what, key, the product's host value, this box's libm value through ctypes, the committed value of the build container)
    for every entry of tests/golden/libm_values.json (lpc.c:192,199,421)"""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "libm_values.json")))
    m = C.CDLL("libm.so.6")
    m.pow.restype = C.c_double; m.pow.argtypes = [C.c_double, C.c_double]
    m.sin.restype = C.c_double; m.sin.argtypes = [C.c_double]
    lib = linne_amd.lib
    lib.lnn_welch_divisor.restype = C.c_double; lib.lnn_welch_divisor.argtypes = [C.c_uint32]
    lib.lnn_sin_window.restype = C.c_double; lib.lnn_sin_window.argtypes = [C.c_uint32, C.c_uint32]
    lib.lnn_cholesky_pivot.restype = C.c_double; lib.lnn_cholesky_pivot.argtypes = [C.c_double]
    for k, v in gold["welch_divisor"].items():
        yield "Welch divisor 4 pow(n - 1, -2), n =", k, lib.lnn_welch_divisor(int(k)), 4.0 * m.pow(float(int(k) - 1), -2.0), float.fromhex(v)
    for k, v in gold["sin_window"].items():
        s, n = (int(t) for t in k.split("/"))
        yield "SIN window sin(pi s / (n - 1)), s/n =", k, lib.lnn_sin_window(s, n), m.sin((3.1415926535897932384626433832795029 * s) / (n - 1)), float.fromhex(v)
    for k, v in gold["cholesky_pivot"].items():
        x = float.fromhex(k)
        yield "Cholesky pivot pow(x, -0.5), x =", k, lib.lnn_cholesky_pivot(x), m.pow(x, -0.5), float.fromhex(v)


def test_host_libm_values_are_the_build_containers():
    """The path takes its transcendental values from the host's libm (lpc.c:192,199,421).  If this box's libm -- or the way the
    library's host code was compiled -- gave other bits than the container the golden streams were made in, every stream hash would
    differ and nothing would say why: this test names the value."""
    n = 0
    for what, key, product, box, committed in libm_values_of_this_box():
        assert box.hex() == committed.hex(), f"this box's libm differs from the build container's: {what} {key}: {box.hex()} vs {committed.hex()}"
        assert product.hex() == committed.hex(), f"liblinne_amd's host value differs: {what} {key}: {product.hex()} vs {committed.hex()}"
        n += 1
    assert n > 100
