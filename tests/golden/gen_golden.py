"""Generate golden fixtures by executing the REFERENCE's own glue from /root/reference under dependency
shims (cv2, easyocr, paddleocr, openai, supervision, torchvision, ultralytics are absent in this image).

Run here (the container that has /root/reference); the fixtures are committed and consumed by tests on
any box.  Only drawing / OCR / model-loading calls are stubbed; `remove_overlap_new`, `int_box_area`,
`get_som_labeled_img` ordering, `YOLOv9Detector._normalize_image_size/_preprocess/_decode/predict` run
from the reference source.  torchvision.ops.batched_nms is shimmed with oracle.detector_ref.batched_nms
(restated from torchvision's CPU source), cv2.resize with oracle.preprocess_ref.cv2_resize_linear.
"""
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))


def install_shims():
    from oracle import detector_ref as D
    from oracle import preprocess_ref as PR

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return self
        def __getattr__(self, n): return _Any()

    cv2 = mod("cv2", resize=lambda img, dsize, **k: PR.cv2_resize_linear(np.asarray(img), dsize),
              rectangle=lambda *a, **k: None, putText=lambda *a, **k: None, getTextSize=lambda *a, **k: ((10, 10), 2),
              cvtColor=lambda img, code: img, FONT_HERSHEY_SIMPLEX=0, LINE_AA=16, FILLED=-1, COLOR_RGB2BGR=4, COLOR_BGR2RGB=4)
    mod("easyocr", Reader=_Any)
    mod("paddleocr", PaddleOCR=_Any)
    mod("openai", AzureOpenAI=_Any)
    mod("ultralytics", YOLO=_Any)
    sv = mod("supervision")
    sv.Detections = _Any
    mod("supervision.detection"); mod("supervision.detection.core", Detections=_Any)
    mod("supervision.draw"); mod("supervision.draw.color", Color=_Any, ColorPalette=_Any)

    def box_convert(boxes, in_fmt, out_fmt):
        if (in_fmt, out_fmt) == ("xyxy", "cxcywh"):
            x1, y1, x2, y2 = boxes.unbind(-1)
            return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), -1)
        cx, cy, w, h = boxes.unbind(-1)
        if (in_fmt, out_fmt) == ("cxcywh", "xyxy"):
            return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1)
        if (in_fmt, out_fmt) == ("cxcywh", "xywh"):
            return torch.stack((cx - 0.5 * w, cy - 0.5 * h, w, h), -1)
        raise NotImplementedError((in_fmt, out_fmt))

    class _NoDraw:      # ref:util/box_annotator.py is cv2 drawing only (visual output, not parity-gated)
        def __init__(self, *a, **k): pass
        def annotate(self, scene, detections, labels=None, skip_label=False, image_size=None): return scene
    pkg = types.ModuleType("util.box_annotator"); pkg.BoxAnnotator = _NoDraw
    sys.modules["util.box_annotator"] = pkg

    tv = mod("torchvision")
    tv.ops = mod("torchvision.ops", box_convert=box_convert, batched_nms=D.batched_nms)
    from PIL import Image
    tv.transforms = mod("torchvision.transforms", ToPILImage=lambda: (lambda a: Image.fromarray(a)), Compose=_Any,
                        RandomResize=_Any, ToTensor=_Any, Normalize=_Any)
    sys.path.insert(0, str(REF))


def rand_boxes(rng, n, w, h, smin, smax):
    xy = rng.uniform(0, 1, size=(n, 2)) * [w - smax, h - smax]
    wh = rng.uniform(smin, smax, size=(n, 2))
    return np.concatenate([xy, xy + wh], 1)


def main():
    install_shims()
    import util.utils as RU          # the reference's own module
    import util.yolov9 as RY
    out = {}
    rng = np.random.default_rng(0)
    # ---- remove_overlap_new / int_box_area
    cases = []
    for ci in range(36):
        w, h = (1920, 1080) if ci % 2 == 0 else (1280, 800)
        n_icon, n_ocr = int(rng.integers(5, 60)), int(rng.integers(0, 25))
        icons = rand_boxes(rng, n_icon, w, h, 10, 120)
        # make nested / overlapping structure
        for j in range(0, n_icon - 1, 3):
            icons[j + 1] = icons[j] + [4, 4, -4, -4] if rng.random() < 0.5 else icons[j] + rng.uniform(-3, 3, 4)
        ocr = rand_boxes(rng, n_ocr, w, h, 8, 200).round()
        for j in range(0, min(n_ocr, n_icon), 4):      # some text inside icons, some icons inside text
            ic = icons[j]
            ocr[j] = [ic[0] + 2, ic[1] + 2, ic[2] - 2, ic[3] - 2] if j % 8 == 0 else [ic[0] - 30, ic[1] - 30, ic[2] + 30, ic[3] + 30]
        if ci % 6 == 3 and n_ocr > 2:
            ocr[1] = ocr[0]              # duplicate OCR element: exercises list.remove of equal dicts
            if ci > 12 and n_ocr > 6:
                ocr[5] = ocr[4]; ocr[6] = ocr[4]
        if ci % 6 == 5 and n_icon > 8:
            icons[7] = icons[6]          # identical icon boxes (equal area: neither is "the larger")
            icons[8, 2:] = icons[8, :2]  # a zero-area icon
        icons_r = (torch.tensor(icons, dtype=torch.float32) / torch.Tensor([w, h, w, h])).tolist()
        ocr_r = (torch.tensor(ocr.astype(np.int64)) / torch.Tensor([w, h, w, h])).tolist() if n_ocr else []
        texts = [f"t{j % 7}" if ci % 6 == 3 else f"t{j}" for j in range(n_ocr)]
        thr = [0.7, 0.9, 0.1][ci % 3]
        ocr_el = [{"type": "text", "bbox": b, "interactivity": False, "content": t, "source": "box_ocr_content_ocr"}
                  for b, t in zip(ocr_r, texts) if RU.int_box_area(b, w, h) > 0]
        icon_el = [{"type": "icon", "bbox": b, "interactivity": True, "content": None} for b in icons_r if RU.int_box_area(b, w, h) > 0]
        import copy
        res = RU.remove_overlap_new(boxes=copy.deepcopy(icon_el), iou_threshold=thr, ocr_bbox=copy.deepcopy(ocr_el) if ocr_el else None)
        cases.append({"w": w, "h": h, "thr": thr, "icons": icon_el, "ocr": ocr_el, "areas": [RU.int_box_area(b, w, h) for b in icons_r],
                      "icons_raw": icons_r, "out": res})
    out["remove_overlap_new"] = cases
    # ---- YOLOv9Detector static pieces
    geo = []
    for (iw, ih, sz) in [(1920, 1080, 640), (1919, 1079, 640), (3240, 2160, 640), (1280, 800, (800, 1280)), (1920, 1080, (1080, 1920)),
                         (300, 900, 640), (641, 641, 640)]:
        tw, th = RY.YOLOv9Detector._normalize_image_size(sz)
        scale = min(tw / iw, th / ih)
        geo.append({"iw": iw, "ih": ih, "imgsz": list(sz) if isinstance(sz, tuple) else sz, "tw": tw, "th": th, "scale": scale,
                    "rw": int(iw * scale), "rh": int(ih * scale)})
    out["geometry"] = geo
    # ---- reference predict() on a tiny TorchScript stand-in vs oracle predict (same blob)
    from oracle import detector_ref as D
    from tools.make_weights import ensure_blob
    from omniparser_amd.synth import synthetic_screenshot, synthetic_ocr
    from PIL import Image
    blob = ensure_blob(seed=1, nc=2, width=0.25)
    det = RY.YOLOv9Detector(model_path=str(blob), device="cpu")
    cpu_model = torch.jit.load(str(blob), map_location="cpu").eval()
    preds = []
    for s, (iw, ih, sz) in enumerate([(640, 480, 320), (1919, 1079, 640), (800, 600, (600, 800))]):
        img = Image.fromarray(synthetic_screenshot(s, iw, ih))
        r = det.predict(img, conf=0.05, imgsz=sz, iou=0.1)[0]
        ob, osc, oc = D.predict(cpu_model, img, conf=0.05, imgsz=sz, iou=0.1)
        assert torch.equal(r.boxes.xyxy, ob) and torch.equal(r.boxes.conf, osc), "oracle.detector_ref.predict != reference predict"
        preds.append({"iw": iw, "ih": ih, "imgsz": list(sz) if isinstance(sz, tuple) else sz, "seed": s,
                      "boxes": r.boxes.xyxy.tolist(), "conf": r.boxes.conf.tolist()})
    out["reference_predict_quarter_width"] = preds
    # ---- get_som_labeled_img ordering with duck-typed models
    class FakeBoxes:
        def __init__(self, xyxy, conf): self.xyxy, self.conf = xyxy, conf
    class FakeDet:
        def __init__(self, xyxy): self.x = xyxy
        def predict(self, source, conf, iou, imgsz=None): return [types.SimpleNamespace(boxes=FakeBoxes(self.x, torch.ones(len(self.x))))]
    class FakeCfg: name_or_path = "florence-fake"; model_type = "florence2"
    class FakeCap:
        config = FakeCfg(); device = torch.device("cpu")
        def generate(self, input_ids=None, pixel_values=None, **k): return torch.arange(pixel_values.shape[0]).view(-1, 1)
    class FakeProc:
        def __call__(self, images=None, text=None, return_tensors="pt", **k):
            b = types.SimpleNamespace(); n = len(images)
            d = {"input_ids": torch.zeros(n, 1, dtype=torch.long), "pixel_values": torch.zeros(n, 3, 8, 8)}
            class B(dict):
                def to(self, **k): return self
            return B(d)
        def batch_decode(self, ids, skip_special_tokens=True): return [f" cap{int(i)} " for i in ids.view(-1)]
    som = []
    for ci, (w, h) in enumerate([(1920, 1080), (1280, 800), (1919, 1079)]):
        img = Image.fromarray(synthetic_screenshot(10 + ci, w, h))
        texts, obox = synthetic_ocr(ci, w, h, 24)
        r2 = np.random.default_rng(100 + ci)
        px = rand_boxes(r2, 40, w, h, 12, 90)
        for j in range(0, 24, 5):
            px[j] = [obox[j][0] - 6, obox[j][1] - 6, obox[j][2] + 6, obox[j][3] + 6]     # icon around a text box
        px[3] = [obox[2][0] + 2, obox[2][1] + 2, obox[2][0] + 8, obox[2][1] + 8]          # icon inside a text box
        px[7] = [10.2, 10.3, 10.9, 30.0]                                                  # zero integer-pixel area
        xyxy = torch.tensor(px, dtype=torch.float32)
        enc, lab, elems = RU.get_som_labeled_img(img, FakeDet(xyxy), BOX_TRESHOLD=0.05, output_coord_in_ratio=True, ocr_bbox=obox,
                                                 draw_bbox_config=None, caption_model_processor={"model": FakeCap(), "processor": FakeProc()},
                                                 ocr_text=texts, use_local_semantics=True, iou_threshold=0.7, scale_img=False, batch_size=16)
        som.append({"w": w, "h": h, "seed": 10 + ci, "ocr_seed": ci, "xyxy": xyxy.tolist(), "elems": elems,
                    "label_keys": list(lab.keys())})
    out["get_som_labeled_img"] = som
    # ---- crop geometry of get_parsed_content_icon (ref:util/utils.py:95-102): `int(coord * W)` multiplies an f32 TENSOR element by a
    # Python int, i.e. in f32 — near-integer pixel coordinates truncate differently than in Python-float arithmetic
    import cv2 as cv2shim
    crops = []
    real_resize = cv2shim.resize
    for ci, (w, h) in enumerate([(1920, 1080), (1919, 1079), (1280, 800)]):
        r3 = np.random.default_rng(500 + ci)
        px = rand_boxes(r3, 48, w, h, 6, 120)
        px[::3] = np.round(px[::3])                                  # exact integer pixel corners
        px[1::6] = np.round(px[1::6]) + r3.uniform(-2e-4, 2e-4, size=px[1::6].shape)
        px[5] = [40.0, 50.0, 40.0, 90.0]                             # empty crop: cv2.resize raises, the reference skips it
        px[9] = [w - 30.5, h - 20.25, w, h]                          # touches the image border
        ratio = torch.tensor(px, dtype=torch.float32) / torch.Tensor([w, h, w, h])
        shapes = []

        def rec_resize(img, dsize, **k):
            if img.shape[0] == 0 or img.shape[1] == 0:
                raise ValueError("empty")
            shapes.append([int(img.shape[0]), int(img.shape[1])])
            return real_resize(img, dsize, **k)
        RU.cv2.resize = rec_resize
        img = np.zeros((h, w, 3), dtype=np.uint8)
        for start in (0, 7):
            shapes.clear()
            RU.get_parsed_content_icon(ratio, start, img, {"model": FakeCap(), "processor": FakeProc()}, batch_size=64)
            crops.append({"w": w, "h": h, "start": start, "ratio": ratio.tolist(), "shapes": [list(x) for x in shapes]})
        RU.cv2.resize = real_resize
    out["crop_shapes"] = crops
    (Path(__file__).parent / "reference_glue.json").write_text(json.dumps(out))
    print("wrote", Path(__file__).parent / "reference_glue.json", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
