"""MI355X-native YOLOv9-E detector behind the reference adapter's interface.

Drop-in for ref:util/yolov9.py `YOLOv9Detector` (same constructor arguments, same
`predict(source, conf, imgsz, iou, max_det) -> [Result]`, same error behaviour), but nothing runs in
PyTorch: `predict` uploads the RGB bytes and replays ONE hipGraph that contains the Pillow-exact
Lanczos letterbox, the whole network as MFMA implicit-GEMM kernels, the fused DFL/decode/threshold
kernel and the torchvision-exact NMS.  There is no CPU fallback — without a gfx950 device or
libomni_amd.so the constructor raises.
"""
import math
import os
from pathlib import Path
from typing import Union

import numpy as np
import torch
from PIL import Image

from .. import _lib as L
from ..planner import PlanBuilder
from ..yolo_graph import YoloV9EGraph
from ..yolo_import import import_state_dict, verify_against_blob

DEFAULT_REPO_ID = "microsoft/OmniParser-v2.0"
DEFAULT_MODEL_FILE = "icon_detect_v3/model.pt"


class Boxes:
    """ref:util/yolov9.py:16-19 (+ `cls`, Ultralytics-style, so class-id parity is testable)."""

    def __init__(self, xyxy: torch.Tensor, confidence: torch.Tensor, cls: torch.Tensor = None):
        self.xyxy = xyxy
        self.conf = confidence
        self.cls = cls


class Result:
    def __init__(self, boxes: Boxes):
        self.boxes = boxes


def _precision_from_env(precision):
    precision = precision or os.environ.get("OMNI_PRECISION", "f32")
    if precision not in ("f32", "f16"):
        raise ValueError(f"precision must be 'f32' or 'f16', got {precision}")
    return L.F32 if precision == "f32" else L.F16


_CONV_TUNING = "unset"


def conv_tuning(device=None):
    """The committed tuning table of the detector's split-f16 convolutions (`omniparser_amd/conv_tuning_gfx950.json`, written by
    tools/conv_autotune.py from serialized graph replays on the MI355X): {shape key: [tile code, split-K count]}.  OMNI_CONV_TUNING=0
    turns it off (A/B), a missing file means the launcher's heuristic everywhere.  Tuning changes which K partials are summed in which
    order, nothing else; it is applied to the detector only (the captioner's plans of different capacities stay bit-identical).
    `device`: a cuda device whose architecture is not gfx950 gets the heuristic (the table is a measurement of that chip)."""
    global _CONV_TUNING
    if device is not None and torch.device(device).type == "cuda":
        arch = getattr(torch.cuda.get_device_properties(device), "gcnArchName", "")
        if not arch.startswith("gfx950"):
            return None
    if _CONV_TUNING == "unset":
        import json
        path = Path(__file__).resolve().parents[1] / "conv_tuning_gfx950.json"
        _CONV_TUNING = None
        if os.environ.get("OMNI_CONV_TUNING", "1") != "0" and path.exists():
            _CONV_TUNING = {k: tuple(v) for k, v in json.loads(path.read_text())["choices"].items()}
    return _CONV_TUNING


class _DetectPlan:
    """Everything one (image size, network size, thresholds) configuration needs."""

    def __init__(self, det: "YOLOv9Detector", iw, ih, image_size, conf, iou, max_det, batch=1):
        tw, th = det._normalize_image_size(image_size)
        scale = min(tw / iw, th / ih)
        rw, rh = int(iw * scale), int(ih * scale)
        pad_left, pad_top = (tw - rw) // 2, (th - rh) // 2
        self.geom = (tw, th, scale, rw, rh, pad_left, pad_top)
        pb = PlanBuilder(det.device, det.dtype)
        pb.conv_tuning = conv_tuning(det.device)               # per-shape tile / split-K choices measured on the MI355X (None: the launcher's heuristic)
        self.pb = pb
        self.batch = batch
        V = pb.V
        self.img = pb.raw((batch, ih, iw, 3), torch.uint8)
        need_h, need_v = int(rw != iw), int(rh != ih)
        x = pb.alloc(batch, th, tw, V)
        tmp = pb.raw((ih, rw, 3), torch.uint8) if need_h else None
        xb = xk = yb = yk = None
        kx = ky = 0
        if need_h:
            b, k = L.resample_coeffs(iw, rw, 0)
            xb, xk, kx = pb.upload(torch.from_numpy(b)), pb.upload(torch.from_numpy(k)), k.shape[1]
        if need_v:
            b, k = L.resample_coeffs(ih, rh, 0)
            yb, yk, ky = pb.upload(torch.from_numpy(b)), pb.upload(torch.from_numpy(k)), k.shape[1]
        for bi in range(batch):
            pb.add_op(L.make_op(
                L.OP_LETTERBOX, det.dtype,
                p=[self.img[bi].data_ptr(), tmp.data_ptr() if tmp is not None else None,
                   xb.data_ptr() if xb is not None else None, xk.data_ptr() if xk is not None else None,
                   x.ptr, yb.data_ptr() if yb is not None else None, yk.data_ptr() if yk is not None else None],
                i={0: ih, 1: iw, 2: rh, 3: rw, 4: kx, 5: ky, 6: th, 7: tw, 8: pad_left, 9: pad_top,
                   10: need_h, 11: need_v, 12: bi, 13: V}))
        n_pre = len(pb.ops)
        graph = YoloV9EGraph(det.state_dict, pb, batch, th, tw, wcache=det._wcache)
        heads = graph.build(x)
        self.heads = heads
        self.debug = graph.debug
        self.x = x
        self.net_ops = (n_pre, len(pb.ops))
        self.net_flops = pb.flops
        self.net_bytes = pb.bytes
        nc = graph.nc
        A = sum((th // s) * (tw // s) for s in (8, 16, 32))
        self.A = A
        esz = 4 if det.dtype == L.F32 else 2
        self.cand = pb.raw((batch, A * L.CAND_BYTES), torch.uint8)
        self.count = pb.raw((batch,), torch.int32)
        self.sorted = pb.raw((batch, (A + 1) * L.CAND_BYTES), torch.uint8)
        nwords = (A + 63) // 64
        self.mask = pb.raw((A * nwords,), torch.int64, zero=False)   # shared scratch: images run back-to-back
        self.out_boxes = pb.raw((batch, max_det, 4), torch.float32)
        self.out_scores = pb.raw((batch, max_det), torch.float32)
        self.out_cls = pb.raw((batch, max_det), torch.int32)
        self.out_count = pb.raw((batch,), torch.int32)
        for bi in range(batch):
            ptrs, ldc, ldb, coffc, coffb = [], [], [], [], []
            for (cls, box) in heads:
                hw = cls.H * cls.W
                ptrs.append((cls.ptr + bi * hw * cls.ld * esz, box.ptr + bi * hw * box.ld * esz))
                ldc.append(cls.ld); ldb.append(box.ld); coffc.append(cls.coff); coffb.append(box.coff)
            pb.add_op(L.make_op(
                L.OP_DETECT_DECODE, det.dtype,
                p=[ptrs[0][0], ptrs[1][0], ptrs[2][0], ptrs[0][1], ptrs[1][1], ptrs[2][1],
                   self.cand[bi].data_ptr(), self.count[bi:].data_ptr()],
                i={0: nc, 1: th, 2: tw, 3: ldc[0], 4: ldc[1], 5: ldc[2], 6: ldb[0], 7: ldb[1], 8: ldb[2], 9: A,
                   10: pad_left, 11: pad_top, 12: 0, 13: coffc[0], 14: coffc[1], 15: coffc[2],
                   16: coffb[0], 17: coffb[1], 18: coffb[2], 19: bi},
                f={0: conf, 1: scale}))
        # ONE NMS op for the whole batch (frames are contiguous in every buffer): up to 2048 candidates per frame — every 640x640 input —
        # one workgroup per frame sorts and suppresses out of LDS, all frames of the batch at once; frames with more candidates
        # (1088x1920 inputs) fall to the tiled kernels inside the same op, one after another over the shared mask scratch
        pb.add_op(L.make_op(
            L.OP_NMS, det.dtype,
            p=[self.cand.data_ptr(), self.count.data_ptr(), self.sorted.data_ptr(), self.mask.data_ptr(), self.out_boxes.data_ptr(),
               self.out_scores.data_ptr(), self.out_cls.data_ptr(), self.out_count.data_ptr()],
            i={0: A, 1: max_det, 2: iw, 3: ih, 4: batch}, f={0: iou}))
        self.plan = pb.build()
        self.n_ops = len(pb.ops)
        torch.cuda.synchronize(det.device)          # buffers were allocated / zero-filled on the current stream: order them before det.stream
        if det.use_graph:
            with torch.cuda.device(det.device):
                self.plan.run(det.stream)            # warm-up (module load) outside capture
                det.stream.synchronize()
                self.plan.capture(det.stream)

    def launch(self, det):
        if det.use_graph:
            self.plan.replay(det.stream)
        else:
            self.plan.run(det.stream)


class YOLOv9Detector:
    """gfx950 YOLOv9-E detector with the reference's Ultralytics-compatible predict API."""

    strides = (8, 16, 32)

    def __init__(
        self,
        model_path: Union[str, Path, None] = None,
        device: Union[str, torch.device, None] = None,
        repo_id: str = DEFAULT_REPO_ID,
        revision: str = "main",
        precision: str = None,
    ):
        self.device = L.require_device(device, "YOLOv9Detector")
        L.lib()   # fail loudly if the HIP extension is missing
        if model_path is None:
            from huggingface_hub import hf_hub_download
            model_path = hf_hub_download(repo_id=repo_id, filename=DEFAULT_MODEL_FILE, revision=revision)
        self.model_path = Path(model_path)
        # The blob is a weight container: its tensors are assigned to their roles from its own graph (yolo_import.py), whatever
        # its attribute names are, and the lowered network is proven against the blob once, at load time, on a probe input.
        blob = torch.jit.load(str(self.model_path), map_location="cpu").eval()
        self.state_dict = import_state_dict(blob)
        self.dtype = _precision_from_env(precision)
        self.use_graph = os.environ.get("OMNI_HIPGRAPH", "1") != "0"
        self.stream = torch.cuda.Stream(device=self.device)
        self._wcache = {}
        self._plans = {}
        self._lock = L.DeviceLock(self.device)   # one inference at a time per detector; makes this GPU the thread's current device
        self.model = self   # callers touch `.model` only to move devices
        self.import_rel_err = None      # largest relative head difference blob vs lowered network on the probe (None = check skipped)
        if os.environ.get("OMNI_VERIFY_IMPORT", "1") != "0":
            nc = self.state_dict["head.cv3.0.2.weight"].shape[0]
            # the probe always runs an f32 plan (the tensor-role mapping it proves does not depend on the plan precision, and f16
            # rounding through ~300 layers would need a tolerance loose enough to hide two similarly scaled sibling tensors)
            self.import_rel_err = verify_against_blob(blob, self._probe_network, nc, tol=2e-2)
        del blob

    def _probe_network(self, x_nchw: torch.Tensor):
        """network part of a plan (no letterbox / decode) on a given input: used once, by the load-time import check."""
        size = x_nchw.shape[-1]
        with torch.cuda.device(self.device):
            pb = PlanBuilder(self.device, L.F32)
            x = pb.alloc(1, size, size, pb.V, zero=True)
            x.t[0, :, :, :3] = x_nchw[0].permute(1, 2, 0).to(x.t.dtype)
            wcache = self._wcache if self.dtype == L.F32 else {}          # f16 detectors: the f32 weight copies live for the probe only
            heads = YoloV9EGraph(self.state_dict, pb, 1, size, size, wcache=wcache).build(x)
            plan = pb.build()
            torch.cuda.synchronize(self.device)
            plan.run(self.stream)
            self.stream.synchronize()
            return [(c.torch().cpu(), b.torch().cpu()) for c, b in heads]

    def to(self, device):   # ref:eval/ss_pro_gpt4o_omniv2.py:30 calls som_model.to(device)
        return self

    @staticmethod
    def _normalize_image_size(image_size):
        if isinstance(image_size, int):
            width = height = image_size
        elif len(image_size) == 2:
            height, width = image_size
        else:
            raise ValueError(f"Expected one or two image dimensions, got {image_size}")
        return ((int(width) + 31) // 32) * 32, ((int(height) + 31) // 32) * 32

    @staticmethod
    def _load_image(source):
        if isinstance(source, Image.Image):
            return source.convert("RGB")
        if isinstance(source, np.ndarray):
            return Image.fromarray(source).convert("RGB")
        with Image.open(source) as image:
            return image.convert("RGB")

    @torch.inference_mode()
    def get_plan(self, iw, ih, imgsz, conf, iou, max_det, batch=1) -> _DetectPlan:
        key = (iw, ih, self._normalize_image_size(imgsz), float(conf), float(iou), int(max_det), batch)
        if key in self._plans:
            self._plans[key] = self._plans.pop(key)          # most recently used last
            return self._plans[key]
        max_plans = int(os.environ.get("OMNI_MAX_DETECT_PLANS", "8"))
        while len(self._plans) >= max_plans:                 # streams of mixed resolutions: bound the buffer pool (LRU)
            self.stream.synchronize()                        # the evicted plan's buffers may still be in use on our stream
            self._plans.pop(next(iter(self._plans)))
            L.gc_after_eviction()
        with torch.cuda.device(self.device):
            self._plans[key] = _DetectPlan(self, iw, ih, imgsz, conf, iou, max_det, batch)
        return self._plans[key]

    @torch.inference_mode()
    def predict_batch(self, images_u8, conf=0.25, imgsz=640, iou=0.7, max_det=300):
        """images_u8: list of equally sized uint8 [H,W,3] arrays/tensors.  Returns list of Result."""
        ih, iw = images_u8[0].shape[:2]
        with self._lock:
            return self._predict_locked(images_u8, iw, ih, conf, imgsz, iou, max_det)

    def _predict_locked(self, images_u8, iw, ih, conf, imgsz, iou, max_det):
        dp = self.get_plan(iw, ih, imgsz, conf, iou, max_det, batch=len(images_u8))
        self.stream.wait_stream(torch.cuda.current_stream(self.device))    # device tensors the caller is still producing on its own stream
        with torch.cuda.stream(self.stream):
            for bi, im in enumerate(images_u8):
                t = im if isinstance(im, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(im))
                dp.img[bi].copy_(t, non_blocking=True)
            dp.launch(self)
            counts = dp.out_count.cpu()    # synchronises the stream
            out = []
            for bi in range(len(images_u8)):
                k = int(counts[bi])
                out.append(Result(Boxes(dp.out_boxes[bi, :k].clone(), dp.out_scores[bi, :k].clone(),
                                        dp.out_cls[bi, :k].clone().long())))
            self.stream.synchronize()      # the clones run on this private stream: finish them before any other stream reads them
        return out

    @torch.inference_mode()
    def predict(self, source, conf=0.25, imgsz=640, iou=0.7, max_det=300):
        image = self._load_image(source)
        arr = np.asarray(image, dtype=np.uint8)
        return self.predict_batch([arr], conf, imgsz, iou, max_det)
