"""Glue parity (tier G of SURVEY 7.5) against fixtures produced by the REFERENCE's own source executed
under dependency shims (tests/golden/gen_golden.py -> tests/golden/reference_glue.json)."""
import copy
import json
import types
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_glue.json").read_text())


def test_remove_overlap_new_matches_reference():
    from omniparser_amd.util.utils import int_box_area, remove_overlap_new
    for case in GOLD["remove_overlap_new"]:
        w, h = case["w"], case["h"]
        assert [int_box_area(b, w, h) for b in case["icons_raw"]] == case["areas"]
        got = remove_overlap_new(copy.deepcopy(case["icons"]), case["thr"], copy.deepcopy(case["ocr"]) if case["ocr"] else None)
        assert got == case["out"]


def test_letterbox_geometry_matches_reference():
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from oracle import detector_ref as D
    for g in GOLD["geometry"]:
        sz = tuple(g["imgsz"]) if isinstance(g["imgsz"], list) else g["imgsz"]
        assert YOLOv9Detector._normalize_image_size(sz) == (g["tw"], g["th"])
        tw, th, scale, rw, rh, pl, pt = D.letterbox_geometry(g["iw"], g["ih"], sz)
        assert (tw, th, scale, rw, rh) == (g["tw"], g["th"], g["scale"], g["rw"], g["rh"])
    with pytest.raises(ValueError):
        YOLOv9Detector._normalize_image_size((1, 2, 3))


def test_oracle_predict_equals_reference_predict():
    """gen_golden.py asserts bitwise equality of oracle.detector_ref.predict with the reference's
    YOLOv9Detector.predict at generation time; here the oracle must still reproduce those boxes."""
    from PIL import Image
    from oracle import detector_ref as D
    from omniparser_amd.synth import synthetic_screenshot
    from tools.make_weights import ensure_blob
    model = torch.jit.load(str(ensure_blob(seed=1, nc=2, width=0.25)), map_location="cpu").eval()
    for rec in GOLD["reference_predict_quarter_width"]:
        sz = tuple(rec["imgsz"]) if isinstance(rec["imgsz"], list) else rec["imgsz"]
        img = Image.fromarray(synthetic_screenshot(rec["seed"], rec["iw"], rec["ih"]))
        b, s, c = D.predict(model, img, conf=0.05, imgsz=sz, iou=0.1)
        ref = torch.tensor(rec["boxes"]).reshape(-1, 4)
        assert len(b) == len(ref)
        if len(ref):
            assert (b - ref).abs().max() < 0.5      # weights are regenerated from the seed on this machine


class _FakeDet:
    def __init__(self, xyxy): self.x = xyxy
    def predict(self, source, conf, iou, imgsz=None):
        return [types.SimpleNamespace(boxes=types.SimpleNamespace(xyxy=self.x, conf=torch.ones(len(self.x))))]


class _FakeCap:
    config = types.SimpleNamespace(name_or_path="florence-fake", model_type="florence2")
    device = torch.device("cpu")
    def caption_crops(self, image, boxes, max_new_tokens=20, batch_size=128):
        return torch.cat([torch.arange(len(boxes[s:s + batch_size])) for s in range(0, len(boxes), batch_size)]).view(-1, 1)


class _FakeProc:
    def batch_decode(self, ids, skip_special_tokens=True): return [f" cap{int(i)} " for i in ids.view(-1)]


def test_get_som_labeled_img_ordering_matches_reference():
    from PIL import Image
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.utils import get_som_labeled_img
    for rec in GOLD["get_som_labeled_img"]:
        w, h = rec["w"], rec["h"]
        img = Image.fromarray(synthetic_screenshot(rec["seed"], w, h))
        texts, obox = synthetic_ocr(rec["ocr_seed"], w, h, 24)
        xyxy = torch.tensor(rec["xyxy"], dtype=torch.float32)
        enc, lab, elems = get_som_labeled_img(img, _FakeDet(xyxy), BOX_TRESHOLD=0.05, output_coord_in_ratio=True, ocr_bbox=obox,
                                              caption_model_processor={"model": _FakeCap(), "processor": _FakeProc()},
                                              ocr_text=texts, use_local_semantics=True, iou_threshold=0.7, batch_size=16)
        assert elems == rec["elems"]
        assert list(lab.keys()) == rec["label_keys"]
        assert isinstance(enc, str) and len(enc) > 100


def test_empty_ocr_is_tolerated():
    """The reference raises TypeError with no OCR boxes (ref:util/utils.py:437-444); the drop-in must not."""
    from PIL import Image
    from omniparser_amd.util.utils import get_som_labeled_img
    img = Image.fromarray(np.zeros((200, 300, 3), dtype=np.uint8))
    xyxy = torch.tensor([[10., 10., 50., 50.], [100., 100., 160., 150.]])
    enc, lab, elems = get_som_labeled_img(img, _FakeDet(xyxy), ocr_bbox=None, ocr_text=[],
                                          caption_model_processor={"model": _FakeCap(), "processor": _FakeProc()})
    assert [e["content"] for e in elems] == ["cap0", "cap1"] and all(e["type"] == "icon" for e in elems)


def test_vectorised_overlap_removal_equals_simple_restatement():
    """randomised cross-check incl. duplicated OCR entries, nested boxes and the no-OCR branch."""
    from omniparser_amd.util.utils import _remove_overlap_new_simple, remove_overlap_new
    rng = np.random.default_rng(7)
    for trial in range(60):
        w, h = 1920, 1080
        n, m = int(rng.integers(0, 80)), int(rng.integers(0, 30))
        xy = rng.uniform(0, 1, (n, 2)) * [w - 150, h - 150]
        wh = rng.uniform(5, 150, (n, 2))
        ic = np.concatenate([xy, xy + wh], 1)
        for j in range(0, max(n - 1, 0), 3):
            ic[j + 1] = ic[j] + rng.uniform(-6, 6, 4)
        oc = np.concatenate([rng.uniform(0, 1, (m, 2)) * [w - 200, h - 60], np.zeros((m, 2))], 1)
        oc[:, 2:] = oc[:, :2] + rng.uniform(8, 200, (m, 2))
        for j in range(0, min(n, m), 2):
            oc[j] = ic[j] + ([3, 3, -3, -3] if j % 4 == 0 else [-20, -20, 20, 20])
        if m > 3:
            oc[2] = oc[1]
        icons = [{"type": "icon", "bbox": b, "interactivity": True, "content": None}
                 for b in (torch.tensor(ic, dtype=torch.float32).reshape(-1, 4) / torch.Tensor([w, h, w, h])).tolist()]
        ocr = [{"type": "text", "bbox": b, "interactivity": False, "content": f"t{j % 5}", "source": "box_ocr_content_ocr"}
               for j, b in enumerate((torch.tensor(oc, dtype=torch.float32).reshape(-1, 4) / torch.Tensor([w, h, w, h])).tolist())]
        thr = [0.1, 0.7, 0.9][trial % 3]
        a = remove_overlap_new(copy.deepcopy(icons), thr, copy.deepcopy(ocr) if (ocr and trial % 5) else None)
        b = _remove_overlap_new_simple(copy.deepcopy(icons), thr, copy.deepcopy(ocr) if (ocr and trial % 5) else None)
        assert a == b, trial


def test_omniparser_facade_on_fake_adapters(monkeypatch):
    """ref:util/omniparser.py:7-32 contract with the device adapters stubbed out: config keys, base64 in,
    (base64 PNG, element list) out, overlay style arithmetic, OCR provider hand-over."""
    import base64
    import io
    from PIL import Image
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util import omniparser as F
    from omniparser_amd.util import utils as U
    xyxy = torch.tensor([[10., 10., 50., 50.], [100., 100., 160., 150.], [300., 40., 340., 90.]])
    seen = {}
    monkeypatch.setattr(U, "get_yolo_model", lambda model_path, device: seen.setdefault("det", (model_path, device)) and _FakeDet(xyxy))
    monkeypatch.setattr(U, "get_caption_model_processor",
                        lambda model_name, model_name_or_path, device: {"model": _FakeCap(), "processor": _FakeProc()})
    for size in [(1920, 1080), (3840, 2160), (640, 480), (3200, 100)]:
        r = max(size) / 3200
        assert F.overlay_style(size) == {"text_scale": 0.8 * r, "text_thickness": max(int(2 * r), 1),
                                         "text_padding": max(int(3 * r), 1), "thickness": max(int(3 * r), 1)}
    cfg = {"som_model_path": "blob.pt", "caption_model_name": "florence2", "caption_model_path": "dir", "BOX_TRESHOLD": 0.05,
           "ocr_provider": lambda image: synthetic_ocr(3, image.size[0], image.size[1], 6)}
    op = F.Omniparser(cfg)
    assert seen["det"][0] == "blob.pt"
    buf = io.BytesIO()
    Image.fromarray(synthetic_screenshot(1, 640, 400)).save(buf, format="PNG")
    b64 = base64.b64encode(buf.getvalue()).decode("ascii")
    png, elems = op.parse(b64)
    assert Image.open(io.BytesIO(base64.b64decode(png))).size == (640, 400)
    assert sum(e["type"] == "text" for e in elems) >= 1 and sum(e["type"] == "icon" for e in elems) >= 1
    assert all(isinstance(e["content"], str) for e in elems)
    assert op.parse_many([b64, b64])[1][1] == elems
    cfg.pop("ocr_provider")
    _, icons_only = F.Omniparser(cfg).parse(b64)
    assert [e["type"] for e in icons_only] == ["icon"] * 3


def test_check_ocr_box_glue_matches_reference():
    """tests/golden/reference_ocr_glue.json: the reference's own check_ocr_box run with fake engines (gen_ocr_golden.py)."""
    import json
    from pathlib import Path
    import numpy as np
    from PIL import Image
    from omniparser_amd.util import utils as U
    recs = json.loads((Path(__file__).resolve().parent / "golden" / "reference_ocr_glue.json").read_text())
    assert len(recs) >= 6
    saved = dict(U._OCR_ENGINES)
    try:
        for r in recs:
            c, res = r["case"], r["results"]
            seen = {}

            class Reader:
                def readtext(self, image, **kw):
                    seen["easyocr_kwargs"], seen["shape"] = kw, list(image.shape)
                    return [(q, t, s) for q, t, s in res]

            class Paddle:
                def ocr(self, image, cls=False):
                    seen["shape"] = list(image.shape)
                    return [[[q, (t, s)] for q, t, s in res]]
            U.set_ocr_engine(Reader(), Paddle())
            w, h = c["size"]
            img = Image.fromarray(np.zeros((h, w, 4 if c["mode"] == "RGBA" else 3), dtype=np.uint8), c["mode"])
            (text, bb), gf = U.check_ocr_box(img, display_img=c["display_img"], output_bb_format=c["fmt"], goal_filtering="gf",
                                             easyocr_args=c["args"], use_paddleocr=c["paddle"])
            assert text == r["text"] and [list(b) for b in bb] == r["bb"] and gf == r["goal_filtering"], c
            assert seen == r["engine_saw"], c
        # precomputed OCR goes through the same formatting; no engine and no result -> no text boxes
        (t, b), _ = U.check_ocr_box(img, display_img=False, output_bb_format="xywh", ocr_result=(["a"], [[10, 20, 50, 44]]))
        assert t == ["a"] and b == [(10, 20, 40, 24)]
        U._OCR_ENGINES.update(easyocr=None, paddleocr=None)
        (t, b), _ = U.check_ocr_box(img, display_img=False, output_bb_format="xyxy")
        assert t == [] and b == []
    finally:
        U._OCR_ENGINES.update(saved)


def test_crop_rectangles_match_reference_f32_truncation():
    """reference_glue.json::crop_shapes — recorded from the reference's own get_parsed_content_icon (crop shapes seen by cv2.resize)."""
    from omniparser_amd.util.utils import crop_boxes_px
    assert len(GOLD["crop_shapes"]) >= 6
    differs_from_f64 = 0
    for rec in GOLD["crop_shapes"]:
        w, h, start = rec["w"], rec["h"], rec["start"]
        boxes = rec["ratio"][start:] if start else rec["ratio"]
        got = [[b[3] - b[1], b[2] - b[0]] for b in crop_boxes_px(boxes, w, h)]
        assert got == rec["shapes"]
        f64 = [[min(int(c[3] * h), h) - int(c[1] * h), min(int(c[2] * w), w) - int(c[0] * w)] for c in boxes
               if int(c[2] * w) - int(c[0] * w) > 0 and int(c[3] * h) - int(c[1] * h) > 0]
        differs_from_f64 += f64 != rec["shapes"]
    assert differs_from_f64 >= 1          # the fixture does exercise the f32-vs-f64 difference
