"""ctypes binding of libomni_amd.so (include/omni_amd.h).

There is NO CPU fallback: if the shared object is missing, cannot be built, or does not export a
symbol the header declares, importing the product path fails loudly.
"""
import ctypes
from ctypes import c_char_p, c_float, c_int, c_int32, c_void_p, POINTER, Structure

from . import build as _build

# GPU_MAX_HW_QUEUES is deliberately left at the HIP runtime's default (4 hardware queues).  With 8, the two-lane pipelined stream
# (pipeline.py::parse_stream) measured +1 % (694 vs 701 ms per bench step), but one encode lane next to the decode stream measured 1034
# instead of 709 ms — which streams end up sharing a hardware pipe depends on the order in which every stream of the process was created
# (RCCL's included), so the faster setting has a cliff next to it that a multi-GPU run could land on (profiles/r3_probe_*.json).

# op kinds / enums (mirror include/omni_amd.h)
F32, F16 = 0, 1
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
OP_CONV, OP_AVGPOOL2, OP_MAXPOOL, OP_RESIZE_NEAREST, OP_LETTERBOX, OP_DETECT_DECODE, OP_NMS = 1, 2, 3, 4, 5, 6, 7
OP_DWCONV3, OP_LAYERNORM, OP_ATTN_ROWS, OP_CHAN_ATTN, OP_PROJ_PREP, OP_ASSEMBLE = 8, 9, 10, 11, 12, 13
OP_EMBED_STEP, OP_ATTN_DECODE, OP_GREEDY_STEP, OP_CROP_RESIZE, OP_DWCONV3_LN, OP_SPLIT_CONVERT, OP_GLUE = 14, 15, 16, 17, 18, 19, 20
OP_OVERLAY, OP_PNG_PACK, OP_PNG_DEFLATE, OP_MLP_FUSED = 21, 22, 23, 24
CAND_BYTES = 32
ABI_VERSION = 3         # include/omni_amd.h::OMNI_ABI_VERSION — a library built from other headers is refused at load time

EXPORTS = [
    "omni_last_error", "omni_abi_version", "omni_device_count", "omni_op_launch",
    "omni_plan_create", "omni_plan_run", "omni_plan_capture", "omni_plan_replay",
    "omni_plan_num_ops", "omni_plan_destroy", "omni_resample_coeffs", "omni_plan_time", "omni_debug_tile_map",
    "omni_plan_profile",
    "omni_model_load", "omni_model_destroy", "omni_model_int", "omni_model_tensor", "omni_model_run",
    "omni_detector_create", "omni_detector_infer", "omni_captioner_create", "omni_captioner_caption",
    "omni_overflow_count",
]


class OmniOp(Structure):
    _fields_ = [
        ("kind", c_int32),
        ("dtype", c_int32),
        ("p", c_void_p * 8),
        ("i", c_int32 * 32),
        ("f", c_float * 8),
    ]


class OmniError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.ensure_built()
    # torch first: its wheel ships its own libamdhip64.so.7, and the device pointers / streams handed to the library come from that
    # runtime.  Loaded afterwards, libomni_amd.so binds to it by SONAME; loaded BEFORE torch it would pull /opt/rocm's copy into the
    # process as a second HIP runtime, whose launches fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    _lib = bind(path)
    return _lib


def bind(path):
    """dlopen a library that implements include/omni_amd.h and declare its prototypes.  The product binds libomni_amd.so (`lib()`);
    the test suite also binds tests/emu/libomni_emu.so — the same device sources compiled for the host — by patching this module
    from tests/emu/emu_runtime.py; nothing in the product does."""
    L = ctypes.CDLL(str(path))
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise OmniError(f"{path} does not export {missing}")
    L.omni_last_error.restype = c_char_p
    L.omni_abi_version.restype = c_int
    L.omni_device_count.restype = c_int
    L.omni_op_launch.argtypes = [POINTER(OmniOp), c_void_p]
    L.omni_op_launch.restype = c_int
    L.omni_plan_create.argtypes = [POINTER(OmniOp), c_int, POINTER(c_void_p)]
    L.omni_plan_create.restype = c_int
    for name in ("omni_plan_run", "omni_plan_capture", "omni_plan_replay"):
        fn = getattr(L, name)
        fn.argtypes = [c_void_p, c_void_p]
        fn.restype = c_int
    L.omni_plan_num_ops.argtypes = [c_void_p]
    L.omni_plan_num_ops.restype = c_int
    L.omni_plan_destroy.argtypes = [c_void_p]
    L.omni_plan_destroy.restype = None
    L.omni_resample_coeffs.argtypes = [c_int, c_int, c_int, POINTER(c_int32), POINTER(c_int32)]
    L.omni_resample_coeffs.restype = c_int
    L.omni_plan_time.argtypes = [c_void_p, c_void_p, c_int, POINTER(c_float)]
    L.omni_plan_time.restype = c_int
    L.omni_plan_profile.argtypes = [c_void_p, c_void_p, POINTER(c_float)]
    L.omni_plan_profile.restype = c_int
    L.omni_debug_tile_map.argtypes = [c_int, c_int, c_int, ctypes.c_longlong, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                      POINTER(c_int)]
    L.omni_debug_tile_map.restype = c_int
    c_ll = ctypes.c_longlong
    for name in ("omni_model_load", "omni_detector_create", "omni_captioner_create"):
        fn = getattr(L, name)
        fn.argtypes = [c_char_p, POINTER(c_void_p)]
        fn.restype = c_int
    L.omni_model_destroy.argtypes = [c_void_p]
    L.omni_model_destroy.restype = None
    L.omni_model_int.argtypes = [c_void_p, c_char_p, POINTER(c_ll)]
    L.omni_model_int.restype = c_int
    L.omni_model_tensor.argtypes = [c_void_p, c_char_p, POINTER(c_void_p), POINTER(c_ll)]
    L.omni_model_tensor.restype = c_int
    L.omni_model_run.argtypes = [c_void_p, c_char_p]
    L.omni_model_run.restype = c_int
    L.omni_detector_infer.argtypes = [c_void_p, c_void_p, c_int, c_int, POINTER(c_float), POINTER(c_float), POINTER(c_int32), POINTER(c_int32)]
    L.omni_detector_infer.restype = c_int
    L.omni_captioner_caption.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_int32), c_int, POINTER(c_int32)]
    L.omni_captioner_caption.restype = c_int
    L.omni_overflow_count.argtypes = [c_int, POINTER(ctypes.c_ulonglong)]
    L.omni_overflow_count.restype = c_int
    if L.omni_abi_version() != ABI_VERSION:
        raise OmniError(f"ABI version mismatch: library {L.omni_abi_version()}, host code {ABI_VERSION} (rebuild: python -m omniparser_amd.build)")
    return L


def overflow_count(reset=True) -> int:
    """Range guard of the split-f16 formats (include/omni_amd.h::omni_overflow_count): threads that produced a GEMM operand beyond
    the f16 range since the last reset.  Synchronous — call it where the host already waits for results."""
    n = ctypes.c_ulonglong(0)
    rc = lib().omni_overflow_count(1 if reset else 0, ctypes.byref(n))
    if rc:
        raise OmniError(lib().omni_last_error().decode())
    return int(n.value)


def gc_after_eviction():
    """A plan that was evicted from a model's plan cache must actually release its device buffers.  `ScreenParser._settle_gc` parks the
    long-lived plan objects in the GC's permanent generation (`gc.freeze`); an evicted plan that sits in a reference cycle would stay
    there — tens of GB of HBM — for ever.  So every eviction thaws, collects and re-freezes (an eviction already costs a device-wide
    synchronise and a plan build: a full collection is noise next to that).  No-op when nothing was frozen."""
    import gc
    if gc.get_freeze_count():
        gc.unfreeze()
        gc.collect()
        gc.freeze()


def require_device(device, what):
    """The product runs on the MI355X only: anything but an available cuda device raises (no CPU fallback)."""
    import torch
    device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
    if device.type == "cuda" and not torch.cuda.is_available():
        raise RuntimeError(f"CUDA device requested but unavailable: {device}")   # ref:util/yolov9.py:40-41
    if device.type != "cuda":
        raise RuntimeError(f"omniparser_amd {what} is the MI355X path and has no CPU fallback; use the reference implementation on CPU")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class DeviceLock:
    """The launch lock of a model (its plans own their device buffers: one inference at a time per model).  Entering it also makes
    the model's GPU the calling thread's current HIP device: the current device is per thread and every new thread starts on
    device 0, so a web-server worker or a pipeline helper thread driving a model on cuda:k (one process per GPU under torchrun
    addresses its GPU as cuda:LOCAL_RANK) would otherwise launch this library's kernels with the wrong device current.  torch's own
    ops guard themselves from their tensors; the ctypes launches have nothing but this.  The previous device is restored on exit."""

    def __init__(self, device, reentrant=False):
        import threading
        import torch
        self._torch = torch
        self._lock = threading.RLock() if reentrant else threading.Lock()
        self._index = device.index if getattr(device, "type", None) == "cuda" else None
        self._local = threading.local()          # per thread: stack of the devices to restore (the reentrant lock nests)

    def __enter__(self):
        self._lock.acquire()
        prev = None
        if self._index is not None:
            cur = self._torch.cuda.current_device()
            if cur != self._index:
                self._torch.cuda.set_device(self._index)
                prev = cur
        stack = getattr(self._local, "stack", None)
        if stack is None:
            stack = self._local.stack = []
        stack.append(prev)
        return self

    def __exit__(self, *exc):
        prev = self._local.stack.pop()
        try:
            if prev is not None:
                self._torch.cuda.set_device(prev)
        finally:
            self._lock.release()
        return False


def check(rc: int):
    if rc != 0:
        raise OmniError(f"libomni_amd error {rc}: {lib().omni_last_error().decode()}")


def make_op(kind, dtype, p=(), i=None, f=None) -> OmniOp:
    """p: sequence of ints/None (device addresses); i/f: dict slot -> value."""
    op = OmniOp()
    op.kind = kind
    op.dtype = dtype
    for k, v in enumerate(p):
        op.p[k] = v if v else None
    for k, v in (i or {}).items():
        op.i[k] = int(v)
    for k, v in (f or {}).items():
        op.f[k] = float(v)
    return op


def launch(op: OmniOp, stream=None):
    check(lib().omni_op_launch(ctypes.byref(op), _stream_ptr(stream)))


def _stream_ptr(stream):
    if stream is None:
        import torch
        if torch.cuda.is_available():
            return c_void_p(torch.cuda.current_stream().cuda_stream)
        return None
    if isinstance(stream, int):
        return c_void_p(stream)
    return c_void_p(stream.cuda_stream)


class Plan:
    """Immutable op list executed by the C++ plan executor (eager or hipGraph replay)."""

    def __init__(self, ops):
        self.ops = list(ops)
        arr = (OmniOp * len(self.ops))(*self.ops)
        h = c_void_p()
        check(lib().omni_plan_create(arr, len(self.ops), ctypes.byref(h)))
        self._h = h
        self.captured = False

    def run(self, stream=None):
        check(lib().omni_plan_run(self._h, _stream_ptr(stream)))

    def capture(self, stream):
        check(lib().omni_plan_capture(self._h, _stream_ptr(stream)))
        self.captured = True

    def replay(self, stream=None):
        check(lib().omni_plan_replay(self._h, _stream_ptr(stream)))

    def time(self, iters, stream=None) -> float:
        ms = c_float()
        check(lib().omni_plan_time(self._h, _stream_ptr(stream), iters, ctypes.byref(ms)))
        return ms.value

    def profile(self, stream=None):
        """device milliseconds of every op of ONE eager replay, in sequence (HIP events around each op)."""
        arr = (c_float * len(self.ops))()
        check(lib().omni_plan_profile(self._h, _stream_ptr(stream), arr))
        return list(arr)

    def __len__(self):
        return len(self.ops)

    def __del__(self):
        try:
            if self._h:
                lib().omni_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def resample_coeffs(in_size: int, out_size: int, filt: int):
    """(bounds int32[out,2], coef int32[out,ksize]) — Pillow's fixed-point tables."""
    import numpy as np
    k = lib().omni_resample_coeffs(in_size, out_size, filt, None, None)
    if k <= 0:
        check(k)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, k), dtype=np.int32)
    rc = lib().omni_resample_coeffs(
        in_size, out_size, filt,
        bounds.ctypes.data_as(POINTER(c_int32)), coef.ctypes.data_as(POINTER(c_int32)))
    if rc <= 0:
        check(rc)
    return bounds, coef


def tile_map(mtiles: int, ntiles: int, bid: int, xcd_n: int = 1, weight_bytes: int = -1):
    """Host mirror of the GEMM block -> tile permutation: returns (mt, nt, grid, xcd_n_used); mt = nt = -1 for padding blocks."""
    mt, nt, grid, used = c_int(), c_int(), c_int(), c_int()
    check(lib().omni_debug_tile_map(mtiles, ntiles, xcd_n, weight_bytes, bid, ctypes.byref(mt), ctypes.byref(nt), ctypes.byref(grid),
                                    ctypes.byref(used)))
    return mt.value, nt.value, grid.value, used.value


class CModel:
    """A plan bundle loaded through the model-level C entry points (include/omni_amd.h: omni_detector_* / omni_captioner_*) — what a
    C / C++ host does, driven from Python for the tests and INTEGRATION.md's examples.  numpy in, numpy out; no torch involved."""

    def __init__(self, path, kind):
        h = c_void_p()
        fn = {"detector": lib().omni_detector_create, "captioner": lib().omni_captioner_create, "model": lib().omni_model_load}[kind]
        check(fn(str(path).encode(), ctypes.byref(h)))
        self._h, self.kind = h, kind

    def int(self, name):
        v = ctypes.c_longlong()
        check(lib().omni_model_int(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    def infer(self, images_u8):
        """images_u8: numpy uint8 [n, H, W, 3] (host) -> (boxes [n, max_det, 4], scores, classes, counts)."""
        import numpy as np
        n, md = images_u8.shape[0], self.int("max_det")
        im = np.ascontiguousarray(images_u8)
        boxes = np.zeros((n, md, 4), np.float32); scores = np.zeros((n, md), np.float32)
        cls = np.zeros((n, md), np.int32); cnt = np.zeros((n,), np.int32)
        check(lib().omni_detector_infer(self._h, im.ctypes.data_as(c_void_p), n, 0, boxes.ctypes.data_as(POINTER(c_float)),
                                        scores.ctypes.data_as(POINTER(c_float)), cls.ctypes.data_as(POINTER(c_int32)),
                                        cnt.ctypes.data_as(POINTER(c_int32))))
        return boxes, scores, cls, cnt

    def caption(self, image_u8, boxes_px):
        """image_u8: numpy uint8 [H, W, 3]; boxes_px: [n, 4] ints -> ids int32 [n, T]."""
        import numpy as np
        im = np.ascontiguousarray(image_u8)
        bx = np.ascontiguousarray(np.asarray(boxes_px, dtype=np.int32).reshape(-1, 4))
        ids = np.zeros((bx.shape[0], self.int("T")), np.int32)
        check(lib().omni_captioner_caption(self._h, im.ctypes.data_as(c_void_p), 0, im.shape[0], im.shape[1],
                                           bx.ctypes.data_as(POINTER(c_int32)), bx.shape[0], ids.ctypes.data_as(POINTER(c_int32))))
        return ids

    def tensor(self, name):
        """(device pointer, nbytes) of a named tensor of the bundle (omni_model_tensor)."""
        p, n = c_void_p(), ctypes.c_longlong()
        check(lib().omni_model_tensor(self._h, name.encode(), ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def close(self):
        if self._h:
            lib().omni_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
