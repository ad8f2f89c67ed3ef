"""linne_amd -- Python bindings (ctypes) of liblinne_amd.so, the MI355X-native implementation of the LINNE
per-frame prediction path behind LINNE's own C API.

The product is the shared library (linne_amd/csrc, built in-tree as linne_amd/liblinne_amd.so); this module
only loads it and wraps its two interfaces:

* ``api``      -- the 13 LINNE public functions (include/linne_encoder.h, include/linne_decoder.h), host buffers;
* ``Context``  -- the batch C-ABI of include/linne_amd.h on device-resident buffers (torch tensors are used
  only to own HBM and to name the HIP stream).

There is no CPU fallback: importing works without a GPU (so the symbol table can be checked), but any
compute call needs a HIP device, and a missing library raises ImportError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LINNE_AMD_LIB") or os.path.join(_HERE, "liblinne_amd.so")     # (LINNE_AMD_LIB: another build of the library, for A/B runs on one box)

PARAM_WORDS = 160
STAT_WORDS = 8
RICE_PLAN_BYTES = 1040         # include/linne_amd.h: [0] order, [1] host-search flag, [16..] parameters
RICE_PLAN_NBITS = 4            # uint32 at this byte offset of a plan: the channel's whole Rice code in bits
PRM_PREV, PRM_PCOEF, PRM_UNITS, PRM_RSHIFT, PRM_COEF = 0, 2, 4, 7, 10
ST_R0, ST_K1, ST_ZERO, ST_TAIL, ST_BEST, ST_LOSS = 0, 1, 4, 5, 6, 7
PRESET_LAYERS = {0: (2, 32), 1: (2, 32), 2: (4, 64, 8), 3: (4, 64, 8), 4: (4, 64, 8),
                 5: (4, 128, 16), 6: (4, 128, 16), 7: (4, 128, 16)}
PRESET_NUM_REGULARS = {0: 1, 1: 2, 2: 1, 3: 2, 4: 4, 5: 1, 6: 2, 7: 4}

# symbols include/*.h declare (checked by tests/test_abi.py)
API_SYMBOLS = [
    "LINNEEncoder_EncodeHeader", "LINNEEncoder_CalculateWorkSize", "LINNEEncoder_Create", "LINNEEncoder_Destroy",
    "LINNEEncoder_SetEncodeParameter", "LINNEEncoder_EncodeBlock", "LINNEEncoder_EncodeWhole",
    "LINNEDecoder_DecodeHeader", "LINNEDecoder_CalculateWorkSize", "LINNEDecoder_Create", "LINNEDecoder_Destroy",
    "LINNEDecoder_SetHeader", "LINNEDecoder_DecodeBlock", "LINNEDecoder_DecodeWhole",
]
AMD_SYMBOLS = [
    "LINNEAmd_GetDeviceCount", "LINNEAmd_ContextCreate", "LINNEAmd_ContextDestroy", "LINNEAmd_GetLastError",
    "LINNEAmd_ReserveScratch", "LINNEAmd_ScratchBytesPerFrame", "LINNEAmd_SetStream", "LINNEAmd_EncodeFramesDevice", "LINNEAmd_DecodeFramesDevice",
    "LINNEAmd_EncodeFramesHost", "LINNEAmd_DecodeFramesHost", "LINNEAmd_Synchronize", "LINNEAmd_GetLastFallbackCount", "LINNEAmd_GetLastTimingMs",
    "LINNEAmd_GetLastTimingLaunches", "LINNEAmd_EnableTiming", "LINNEAmd_PackFrames",
    "LINNEAmd_GetLastMinMargin", "LINNEAmd_SetAfIterations", "LINNEAmd_SetLearning", "LINNEAmd_MultiCreate", "LINNEAmd_MultiDestroy", "LINNEAmd_MultiNumDevices", "LINNEAmd_MultiDevice",
    "LINNEAmd_MultiContext", "LINNEAmd_MultiGetLastError", "LINNEAmd_MultiEncodeFramesHost", "LINNEAmd_MultiDecodeFramesHost", "LINNEAmd_SlotCreate", "LINNEAmd_SlotDestroy", "LINNEAmd_SlotPcm", "LINNEAmd_SlotData", "LINNEAmd_SlotParams", "LINNEAmd_SlotStats",
    "LINNEAmd_SlotCapacity", "LINNEAmd_SlotRicePlan", "LINNEAmd_SlotCreateEx", "LINNEAmd_SlotFlags", "LINNEAmd_SlotPcm16", "LINNEAmd_SlotPacked", "LINNEAmd_SlotOffsets",
    "LINNEAmd_SlotFetchResidual", "LINNEAmd_SlotStream", "LINNEAmd_SlotStreamCapacity", "LINNEAmd_SlotBitPos", "LINNEAmd_SlotEndBits", "LINNEAmd_SlotPcm16Valid",
    "LINNEAmd_SlotPcmWidth", "LINNEAmd_SlotDecodeStreamSubmit", "LINNEAmd_SlotFetchPcm32", "LINNEAmd_RiceDecodeDevice", "LINNEAmd_SlotBitEnd", "LINNEAmd_LastDecodeWholeMode", "LINNEAmd_RiceEmitDevice", "LINNEAmd_PackFramesEmitted", "LINNEAmd_RicePlanDevice", "LINNEAmd_PackFramesPlanned", "LINNEAmd_SlotEncodeSubmit", "LINNEAmd_SlotDecodeSubmit", "LINNEAmd_SlotWait",
]


class Shape(C.Structure):
    _fields_ = [("num_channels", C.c_uint32), ("bits_per_sample", C.c_uint32), ("num_samples_per_block", C.c_uint32),
                ("preset", C.c_uint32), ("ch_process_method", C.c_uint32)]


def _load():
    # torch bundles its own HIP runtime (same soname as /opt/rocm's).  Importing torch FIRST makes
    # liblinne_amd.so bind to that already-loaded runtime, so both share one HIP context: torch tensors can
    # be handed to the C-ABI and torch.cuda streams/events see our work.  The other order loads two runtimes.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C linne_amd/csrc`). linne_amd has no pure-Python or CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.LINNEAmd_GetDeviceCount.restype = C.c_int
    L.LINNEAmd_ContextCreate.restype = C.c_void_p
    L.LINNEAmd_ContextCreate.argtypes = [C.c_int, C.c_uint64]
    L.LINNEAmd_ContextDestroy.argtypes = [C.c_void_p]
    L.LINNEAmd_GetLastError.restype = C.c_char_p
    L.LINNEAmd_GetLastError.argtypes = [C.c_void_p]
    L.LINNEAmd_ReserveScratch.argtypes = [C.c_void_p, C.c_uint64]
    L.LINNEAmd_ScratchBytesPerFrame.restype = C.c_uint64
    L.LINNEAmd_ScratchBytesPerFrame.argtypes = [C.POINTER(Shape)]
    L.LINNEAmd_SetStream.argtypes = [C.c_void_p, C.c_void_p]
    L.LINNEAmd_EncodeFramesDevice.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
    L.LINNEAmd_DecodeFramesDevice.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.LINNEAmd_EncodeFramesHost.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
    L.LINNEAmd_DecodeFramesHost.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.LINNEAmd_Synchronize.argtypes = [C.c_void_p]
    L.LINNEAmd_GetLastFallbackCount.restype = C.c_int64
    L.LINNEAmd_GetLastFallbackCount.argtypes = [C.c_void_p]
    L.LINNEAmd_SetAfIterations.argtypes = [C.c_void_p, C.c_uint32]
    L.LINNEAmd_SetLearning.argtypes = [C.c_void_p, C.c_uint32]
    L.LINNEAmd_GetLastMinMargin.restype = C.c_double
    L.LINNEAmd_GetLastMinMargin.argtypes = [C.c_void_p]
    L.LINNEAmd_GetLastTimingMs.restype = C.c_double
    L.LINNEAmd_GetLastTimingMs.argtypes = [C.c_void_p, C.c_int]
    L.LINNEAmd_EnableTiming.argtypes = [C.c_void_p, C.c_int]
    L.LINNEAmd_GetLastTimingLaunches.restype = C.c_int
    L.LINNEAmd_GetLastTimingLaunches.argtypes = [C.c_void_p, C.c_int]
    L.LINNEAmd_PackFrames.argtypes = [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_double), C.c_uint32]
    L.LINNEAmd_PackFramesPlanned.argtypes = [C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_double), C.c_uint32]
    L.LINNEAmd_RicePlanDevice.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.LINNEAmd_RiceEmitDevice.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.LINNEAmd_PackFramesEmitted.argtypes = [C.POINTER(Shape), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.POINTER(C.c_double), C.c_uint32]
    L.LINNEAmd_MultiCreate.restype = C.c_void_p
    L.LINNEAmd_MultiCreate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    L.LINNEAmd_MultiDestroy.argtypes = [C.c_void_p]
    L.LINNEAmd_MultiNumDevices.restype = C.c_uint32
    L.LINNEAmd_MultiNumDevices.argtypes = [C.c_void_p]
    L.LINNEAmd_MultiDevice.argtypes = [C.c_void_p, C.c_uint32]
    L.LINNEAmd_MultiContext.restype = C.c_void_p
    L.LINNEAmd_MultiContext.argtypes = [C.c_void_p, C.c_uint32]
    L.LINNEAmd_MultiGetLastError.restype = C.c_char_p
    L.LINNEAmd_MultiGetLastError.argtypes = [C.c_void_p]
    L.LINNEAmd_MultiEncodeFramesHost.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_uint32]
    L.LINNEAmd_MultiDecodeFramesHost.argtypes = [C.c_void_p, C.POINTER(Shape), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    return L


lib = _load()


def device_count():
    return int(lib.LINNEAmd_GetDeviceCount())


class LinneAmdError(RuntimeError):
    pass


class Context:
    """Batch hot path on one GPU (include/linne_amd.h).  Tensors are torch int32/float64 CUDA(HIP) tensors."""

    def __init__(self, device=0, scratch_bytes=0, use_torch_stream=True):
        self.h = lib.LINNEAmd_ContextCreate(int(device), int(scratch_bytes))
        if not self.h:
            raise LinneAmdError(f"LINNEAmd_ContextCreate(device={device}) failed: no usable HIP device "
                                "(the prediction path has no CPU fallback)")
        self.device = int(device)
        # With its own stream the library's kernels are NOT ordered behind what torch enqueued on torch's stream -- the fill kernel of a
        # torch.zeros, a clone, a copy from the host: the binding then drains torch's stream before every call that touches tensors
        # (_fence).  It went unnoticed while both streams happened to share a hardware queue; with the library's 24 queues they do not.
        self._own_stream = not use_torch_stream
        if use_torch_stream:
            import torch
            s = torch.cuda.current_stream(self.device).cuda_stream
            self._check(lib.LINNEAmd_SetStream(self.h, C.c_void_p(s)), "SetStream")

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.LINNEAmd_ContextDestroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, ret, what):
        if ret != 0:
            raise LinneAmdError(f"{what} -> {ret}: {lib.LINNEAmd_GetLastError(self.h).decode()}")

    def _fence(self):
        """own stream: everything torch has enqueued so far (allocations' fills, clones, H2D copies) is done before the library runs"""
        if self._own_stream:
            import torch
            torch.cuda.current_stream(self.device).synchronize()

    @staticmethod
    def shape(nch, bits, block, preset, ms):
        return Shape(nch, bits, block, preset, int(ms))

    def reserve(self, nbytes):
        self._check(lib.LINNEAmd_ReserveScratch(self.h, int(nbytes)), "ReserveScratch")

    def enable_timing(self, on=True):
        lib.LINNEAmd_EnableTiming(self.h, int(on))

    def last_ms(self, which=0):
        return float(lib.LINNEAmd_GetLastTimingMs(self.h, which))

    def last_launches(self, which):
        return int(lib.LINNEAmd_GetLastTimingLaunches(self.h, which))

    def last_fallback_count(self):
        return int(lib.LINNEAmd_GetLastFallbackCount(self.h))

    def set_learning(self, on):
        self._check(lib.LINNEAmd_SetLearning(self.h, int(bool(on))), "SetLearning")

    def set_af_iterations(self, n):
        self._check(lib.LINNEAmd_SetAfIterations(self.h, int(n)), "SetAfIterations")

    def last_min_margin(self):
        return float(lib.LINNEAmd_GetLastMinMargin(self.h))

    def synchronize(self):
        self._check(lib.LINNEAmd_Synchronize(self.h), "Synchronize")

    def encode_frames(self, shape, pcm, num_samples=None, out=None):
        """pcm: int32 cuda tensor [F][C][S] -> (residual [F][C][S] int32, params [F][C][160] int32, stats [F][C][8] f64)"""
        import torch
        F, Cn, S = pcm.shape
        assert pcm.dtype == torch.int32 and pcm.is_cuda and pcm.is_contiguous()
        assert Cn == shape.num_channels and S == shape.num_samples_per_block
        if out is None:
            res = torch.empty_like(pcm)
            prm = torch.zeros((F, Cn, PARAM_WORDS), dtype=torch.int32, device=pcm.device)
            st = torch.zeros((F, Cn, STAT_WORDS), dtype=torch.float64, device=pcm.device)
        else:
            res, prm, st = out
        ns = None
        if num_samples is not None:
            ns = np.ascontiguousarray(num_samples, dtype=np.uint32)
            assert ns.shape == (F,)
        self._fence()
        self._check(lib.LINNEAmd_EncodeFramesDevice(self.h, C.byref(shape), pcm.data_ptr(), ns.ctypes.data if ns is not None else None,
                                                    F, res.data_ptr(), prm.data_ptr(), st.data_ptr()), "EncodeFramesDevice")
        return res, prm, st

    def rice_plan(self, shape, residual, num_samples=None):
        """residual: int32 cuda tensor [F][C][S] -> uint8 cuda tensor [F][C][RICE_PLAN_BYTES] (order, flag, parameters)"""
        import torch
        F, Cn, S = residual.shape
        assert residual.dtype == torch.int32 and residual.is_cuda and residual.is_contiguous()
        plan = torch.zeros((F, Cn, RICE_PLAN_BYTES), dtype=torch.uint8, device=residual.device)
        ns = np.ascontiguousarray(num_samples, dtype=np.uint32) if num_samples is not None else None
        self._fence()
        self._check(lib.LINNEAmd_RicePlanDevice(self.h, C.byref(shape), residual.data_ptr(), ns.ctypes.data if ns is not None else None,
                                                F, plan.data_ptr()), "RicePlanDevice")
        return plan

    def rice_emit(self, shape, residual, plan, capacity_bytes=None):
        """residual int32 cuda [F][C][S], plan = rice_plan(...) of the same batch -> (packed uint8 cuda [capacity], offsets uint32-as-int32
        cuda [F*C+1]): every channel-frame's Rice code, written by the device (include/linne_amd.h LINNEAmd_RiceEmitDevice)"""
        import torch
        F, Cn, S = residual.shape
        cap = int(capacity_bytes if capacity_bytes is not None else F * Cn * (S * 4 + 64))
        packed = torch.zeros(cap, dtype=torch.uint8, device=residual.device)
        offsets = torch.zeros(F * Cn + 1, dtype=torch.int32, device=residual.device)
        self._fence()
        self._check(lib.LINNEAmd_RiceEmitDevice(self.h, C.byref(shape), residual.data_ptr(), F, plan.data_ptr(), offsets.data_ptr(),
                                                packed.data_ptr(), cap), "RiceEmitDevice")
        return packed, offsets

    def decode_frames(self, shape, data, params, num_samples=None):
        """in place: data int32 cuda [F][C][S] residual -> PCM"""
        import torch
        F, Cn, S = data.shape
        assert data.dtype == torch.int32 and data.is_cuda and data.is_contiguous() and params.is_contiguous()
        ns = None
        if num_samples is not None:
            ns = np.ascontiguousarray(num_samples, dtype=np.uint32)
        self._fence()
        self._check(lib.LINNEAmd_DecodeFramesDevice(self.h, C.byref(shape), data.data_ptr(), ns.ctypes.data if ns is not None else None,
                                                    F, params.data_ptr()), "DecodeFramesDevice")
        return data

    def encode_frames_host(self, shape, pcm, num_samples=None):
        """numpy int32 [F][C][S] -> numpy (residual, params, stats)"""
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        F, Cn, S = pcm.shape
        res = np.zeros_like(pcm)
        prm = np.zeros((F, Cn, PARAM_WORDS), dtype=np.int32)
        st = np.zeros((F, Cn, STAT_WORDS), dtype=np.float64)
        ns = np.ascontiguousarray(num_samples, dtype=np.uint32) if num_samples is not None else None
        self._check(lib.LINNEAmd_EncodeFramesHost(self.h, C.byref(shape), pcm.ctypes.data, ns.ctypes.data if ns is not None else None,
                                                  F, res.ctypes.data, prm.ctypes.data, st.ctypes.data), "EncodeFramesHost")
        return res, prm, st

    def decode_frames_host(self, shape, residual, params, num_samples=None):
        d = np.ascontiguousarray(residual, dtype=np.int32).copy()
        prm = np.ascontiguousarray(params, dtype=np.int32)
        ns = np.ascontiguousarray(num_samples, dtype=np.uint32) if num_samples is not None else None
        self._check(lib.LINNEAmd_DecodeFramesHost(self.h, C.byref(shape), d.ctypes.data, ns.ctypes.data if ns is not None else None,
                                                  d.shape[0], prm.ctypes.data), "DecodeFramesHost")
        return d


class Multi:
    """Several GPUs from one process (include/linne_amd.h LINNEAmd_Multi*): groups of frames fan out round-robin over per-GPU
    contexts, each GPU fed over its own PCIe link; host numpy arrays in and out, the caller's frame order kept."""

    def __init__(self, devices=None, scratch_bytes=0):
        if devices:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            self.h = lib.LINNEAmd_MultiCreate(arr, len(devices), int(scratch_bytes))
        else:
            self.h = lib.LINNEAmd_MultiCreate(None, 0, int(scratch_bytes))
        if not self.h:
            raise LinneAmdError(f"LINNEAmd_MultiCreate({devices}) failed: no usable HIP device (there is no CPU fallback)")

    @property
    def num_devices(self):
        return int(lib.LINNEAmd_MultiNumDevices(self.h))

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.LINNEAmd_MultiDestroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, ret, what):
        if ret != 0:
            raise LinneAmdError(f"{what} -> {ret}: {lib.LINNEAmd_MultiGetLastError(self.h).decode()}")

    def encode_frames_host(self, shape, pcm, num_samples=None, group_frames=0, want_plan=False, out=None):
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        F, Cn, S = pcm.shape
        if out is None:
            res = np.empty_like(pcm)
            prm = np.zeros((F, Cn, PARAM_WORDS), dtype=np.int32)
            st = np.zeros((F, Cn, STAT_WORDS), dtype=np.float64)
        else:
            res, prm, st = out
        plan = np.zeros((F, Cn, RICE_PLAN_BYTES), dtype=np.uint8) if want_plan else None
        ns = np.ascontiguousarray(num_samples, dtype=np.uint32) if num_samples is not None else None
        self._check(lib.LINNEAmd_MultiEncodeFramesHost(self.h, C.byref(shape), pcm.ctypes.data, ns.ctypes.data if ns is not None else None, F,
                                                       res.ctypes.data, prm.ctypes.data, st.ctypes.data,
                                                       plan.ctypes.data if plan is not None else None, int(group_frames)), "MultiEncodeFramesHost")
        return (res, prm, st, plan) if want_plan else (res, prm, st)

    def decode_frames_host(self, shape, residual, params, num_samples=None, group_frames=0, in_place=False):
        d = np.ascontiguousarray(residual, dtype=np.int32)
        if not in_place:
            d = d.copy()
        prm = np.ascontiguousarray(params, dtype=np.int32)
        ns = np.ascontiguousarray(num_samples, dtype=np.uint32) if num_samples is not None else None
        self._check(lib.LINNEAmd_MultiDecodeFramesHost(self.h, C.byref(shape), d.ctypes.data, ns.ctypes.data if ns is not None else None,
                                                       d.shape[0], prm.ctypes.data, int(group_frames)), "MultiDecodeFramesHost")
        return d


def pack_frames_emitted(shape, planes, first_sample, params, stats, plan, packed, offsets, num_samples=None, parcor_state=0.0, threads=0, residual=None):
    """host stitch stage over the device's Rice codes (LINNEAmd_PackFramesEmitted): planes = int32 [C][total samples] (the caller's PCM),
    the batch's frames start at first_sample; residual [F][C][S] (numpy) serves the fetch callback for channel-frames without a code"""
    planes = np.ascontiguousarray(planes, dtype=np.int32)
    params = np.ascontiguousarray(params, dtype=np.int32)
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    plan = np.ascontiguousarray(plan, dtype=np.uint8)
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets).view(np.uint32)
    F, Cn = params.shape[0], params.shape[1]
    ptrs = (C.POINTER(C.c_int32) * Cn)(*[planes[ch].ctypes.data_as(C.POINTER(C.c_int32)) for ch in range(Cn)])
    cap = F * Cn * shape.num_samples_per_block * 8 + 64 * F + 64
    out = np.zeros(cap, dtype=np.uint8)
    sizes = np.zeros(F, dtype=np.uint32)
    ns = np.ascontiguousarray(num_samples, dtype=np.uint32) if num_samples is not None else None
    st = C.c_double(parcor_state)
    fetched = []
    FETCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32))

    def _fetch(arg, frame, dst):
        if residual is None:
            return 7
        fetched.append(int(frame))
        C.memmove(dst, np.ascontiguousarray(residual[frame], dtype=np.int32).ctypes.data, Cn * shape.num_samples_per_block * 4)
        return 0

    cb = FETCH(_fetch)
    ret = lib.LINNEAmd_PackFramesEmitted(C.byref(shape), ptrs, int(first_sample), ns.ctypes.data if ns is not None else None, F,
                                         params.ctypes.data, stats.ctypes.data, plan.ctypes.data, packed.ctypes.data, offsets.ctypes.data,
                                         C.cast(cb, C.c_void_p), None, out.ctypes.data, cap, sizes.ctypes.data, C.byref(st), threads or (os.cpu_count() or 1))
    if ret != 0:
        raise LinneAmdError(f"PackFramesEmitted -> {ret}")
    blocks, off = [], 0
    for s in sizes:
        blocks.append(out[off:off + int(s)].tobytes())
        off += int(s)
    return blocks, st.value, fetched


def pack_frames(shape, pcm, residual, params, stats, num_samples=None, parcor_state=0.0, threads=0, plan=None):
    """host entropy stage: numpy arrays of one batch -> (list of block bytes, new parcor_state); plan = the device's
    Rice plan (Context.rice_plan, as a numpy uint8 array) or None for the host search"""
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    residual = np.ascontiguousarray(residual, dtype=np.int32)
    params = np.ascontiguousarray(params, dtype=np.int32)
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    F = pcm.shape[0]
    cap = pcm.size * 8 + 64 * F + 64
    out = np.zeros(cap, dtype=np.uint8)
    sizes = np.zeros(F, dtype=np.uint32)
    ns = np.ascontiguousarray(num_samples, dtype=np.uint32) if num_samples is not None else None
    st = C.c_double(parcor_state)
    if plan is not None:
        plan = np.ascontiguousarray(plan, dtype=np.uint8)
        assert plan.shape == (F, pcm.shape[1], RICE_PLAN_BYTES)
    ret = lib.LINNEAmd_PackFramesPlanned(C.byref(shape), pcm.ctypes.data, ns.ctypes.data if ns is not None else None, F,
                                         residual.ctypes.data, params.ctypes.data, stats.ctypes.data,
                                         plan.ctypes.data if plan is not None else None, out.ctypes.data, cap,
                                         sizes.ctypes.data, C.byref(st), threads or (os.cpu_count() or 1))
    if ret != 0:
        raise LinneAmdError(f"PackFrames -> {ret}")
    blocks, off = [], 0
    for s in sizes:
        blocks.append(out[off:off + int(s)].tobytes())
        off += int(s)
    return blocks, st.value
