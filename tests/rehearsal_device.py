"""CPU rehearsal of the `-m gpu` detector checks (tests/gpu_checks.py::check_detector + assert_detector_frame).

`RehearsalDetector` offers the attributes check_detector reads from `omniparser_amd.util.yolov9.YOLOv9Detector` and its plan,
computed by the ORACLE network evaluated in float64 and rounded to float32: its difference from the float32 oracle is rounding
noise of the order the MI355X path shows (measured GPU-vs-oracle head error = 1.7x ... 3x the f32-vs-f64 difference), its
post-processing is the restated reference code, its candidate records come in scrambled order like the device's atomic compaction.
It is test infrastructure for tests of tests: nothing in the product imports it."""
import types

import torch

from oracle import detector_ref as D
from oracle.yolov9e_ref import YOLOv9E
from omniparser_amd.pipeline import ScreenParser as _RealScreenParser      # bound before any test monkeypatches the module attribute


class _V:
    def __init__(self, t):
        self.t = t

    def torch(self):
        return self.t


class RehearsalDetector:
    def __init__(self, model_path, width, nc=1, device="cuda", precision="f32"):
        cpu_model = torch.jit.load(str(model_path), map_location="cpu").eval()
        self.m64 = YOLOv9E(nc=nc, width=width).double()
        self.m64.load_state_dict({k: v.double() for k, v in cpu_model.state_dict().items()}, strict=False)
        self.m64.eval()
        self.raw_box = {}
        for i, seq in enumerate(self.m64.head.cv2):
            seq.register_forward_hook(lambda mod, inp, out, i=i: self.raw_box.__setitem__(i, out.detach()))
        self.plan = None

    @torch.inference_mode()
    def predict(self, source, conf=0.25, imgsz=640, iou=0.7, max_det=300):
        image = D.load_image(source)
        x, scale, pad_left, pad_top = D.preprocess(image, imgsz)
        outs = [o.float() for o in self.m64(x.double())]
        boxes, scores, cls, dbg = D.postprocess([o.clone() for o in outs], image.width, image.height, scale, pad_left, pad_top, conf, iou,
                                                max_det)
        cb, cs, cc = dbg["cand"]
        anchors = torch.nonzero(dbg["valid"]).flatten()
        n = len(anchors)
        rec = torch.zeros(dbg["valid"].numel(), 8, dtype=torch.int32)
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
        rf = rec.view(torch.float32)
        rf[:n, 0:4] = cb[perm]
        rf[:n, 4] = cs[perm]
        rec[:n, 5] = cc[perm].int()
        rec[:n, 6] = anchors[perm].int()
        xin = torch.zeros(1, x.shape[2], x.shape[3], 4)
        xin[0, :, :, :3] = x[0].permute(1, 2, 0)
        self.plan = types.SimpleNamespace(
            heads=[(_V(outs[2 * i]), _V(self.raw_box[i].float())) for i in range(3)], x=_V(xin), count=torch.tensor([n]),
            cand=rec.view(1, -1, 8), n_ops=0, net_flops=0.0)
        res = types.SimpleNamespace(boxes=types.SimpleNamespace(xyxy=boxes, conf=scores, cls=cls))
        return [res]

    def get_plan(self, *a, **k):
        return self.plan


class RehearsalCaptioner:
    """what check_bench_path reads from Florence2Captioner, backed by the transformers oracle (`oc` = gpu_checks._OracleCaptioner)."""

    def __init__(self, oc, R):
        self.oc, self.R = oc, R
        self.max_new_tokens = 20
        self.w = types.SimpleNamespace(pad=1)
        self.x_in = {}

    @staticmethod
    def bucket(n):
        b = 1
        while b < n:
            b *= 2
        return b

    def plans(self, B, R, max_new):
        return types.SimpleNamespace(x_in=_V(self.x_in[B]))


class RehearsalParser:
    """what check_bench_path reads from pipeline.ScreenParser: parse_batch(frames, ocr, return_ids=True), last_crops, glue, batch_size —
    detector = RehearsalDetector, hand-off = the product's host twin, captions = the oracle, crops packed across frames into
    micro-batches of `batch_size` like the product does."""
    micro_batch = 32

    def __init__(self, det, cap, box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640, batch_size=128):
        self._sp = _RealScreenParser(None, None, processor=object(), box_threshold=box_threshold, iou_threshold=iou_threshold, nms_iou=nms_iou,
                                max_det=max_det, imgsz=imgsz)
        self.det, self.cap, self.glue = det, cap, self._sp.glue
        self.conf, self.nms_iou, self.imgsz, self.batch_size = box_threshold, nms_iou, imgsz, self.micro_batch
        self.last_crops = None

    def parse_batch(self, frames, ocr, return_ids=False):
        import numpy as np
        from PIL import Image
        from oracle import preprocess_ref as PR
        from omniparser_amd.florence import CLIP_MEAN, CLIP_STD
        elems_all, crops_all, ids_out = [], [], []
        imgs = [f.cpu().numpy() for f in frames]
        cands, counts = [], []
        for fi, img in enumerate(imgs):
            ih, iw = img.shape[:2]
            xy = self.det.predict(Image.fromarray(img), conf=self.conf, imgsz=self.imgsz, iou=self.nms_iou)[0].boxes.xyxy
            cands.append(self.det.plan.cand[0]); counts.append(self.det.plan.count[0])
            texts, boxes = ocr[fi]
            el, cr = self.glue(xy, iw, ih, boxes, texts)
            rows = self.cap.oc.caption_crops(img, cr, max_new_tokens=20, batch_size=64) if cr else []
            elems_all.append(el); crops_all.append(cr); ids_out.append([r for r in rows])
        flat = [(f, k) for f in range(len(imgs)) for k in range(len(crops_all[f]))]
        n_last = len(flat) % self.batch_size or min(len(flat), self.batch_size)
        B, R = self.cap.bucket(n_last), self.cap.R
        x = torch.zeros(B, R, R, 4)
        for j, (f, k) in enumerate(flat[len(flat) - n_last:]):
            x[j, :, :, :3] = torch.from_numpy(PR.caption_pixel_values(imgs[f], crops_all[f][k], R, CLIP_MEAN, CLIP_STD))
        self.cap.x_in[B] = x
        self.last_crops = crops_all
        self.det.plan = types.SimpleNamespace(cand=torch.stack(cands), count=torch.stack(counts))      # the batch plan's candidate records
        self._last = (elems_all, ids_out)
        return (elems_all, ids_out) if return_ids else elems_all

    def parse_stream(self, batches, return_ids=False):
        """the rehearsal has one 'device' evaluation per frame: the pipelined composition replays the batch it already parsed (what the
        check compares is the product's pipeline against the product's parse_batch — a statement about HIP streams, not about this stand-in)"""
        for _frames, _ocr in batches:
            yield self._last if return_ids else self._last[0]
