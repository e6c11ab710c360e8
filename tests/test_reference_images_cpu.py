"""The oracle and the host glue on the REFERENCE'S OWN images (tests/golden/ref_imgs: word.png 1919x1079 RGBA, demo_image.jpg
3240x2160 = BASELINE configs[0]) against vectors the reference's own source produced on them (tests/golden/gen_reference_images.py
-> tests/golden/reference_images.json): letterbox geometry + network-input digest, final boxes / scores bit for bit, the element
list and the crop rectangles of get_som_labeled_img."""
import hashlib
import json
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE / "golden"))
GOLD = json.loads((HERE / "golden" / "reference_images.json").read_text())
IMG_DIR = HERE / "golden" / "ref_imgs"


@pytest.fixture(scope="module")
def standin():
    from gen_reference_images import state_digest
    from tools.make_weights import ensure_blob
    blob = ensure_blob(seed=0, nc=1, width=1.0)
    same = state_digest(blob) == GOLD["blob"]["state_sha256"]
    return torch.jit.load(str(blob), map_location="cpu").eval(), same


def test_fixture_files_are_the_reference_images():
    from PIL import Image
    for name, rec in GOLD["images"].items():
        assert hashlib.sha256((IMG_DIR / name).read_bytes()).hexdigest() == rec["file_sha256"]
        img = Image.open(IMG_DIR / name)
        assert list(img.size) == rec["size"] and img.mode == rec["mode"]
    assert GOLD["images"]["word.png"]["size"] == [1919, 1079] and GOLD["images"]["word.png"]["mode"] == "RGBA"
    assert GOLD["images"]["demo_image.jpg"]["size"] == [3240, 2160]


@pytest.mark.parametrize("name", sorted(GOLD["images"]))
def test_oracle_letterbox_equals_reference_on_its_own_images(name):
    """ref:util/yolov9.py:73-87 executed by the reference at generation time: int() truncation of the resized size (1919 x 1079 ->
    640 x 359, not 360), pads, the Lanczos-resampled network input byte for byte (u8 digest) — and no weights involved."""
    from PIL import Image
    from oracle import detector_ref as D
    rec = GOLD["images"][name]["preprocess"]
    img = Image.open(IMG_DIR / name).convert("RGB")
    x, scale, pad_left, pad_top = D.preprocess(img, GOLD["imgsz"])
    assert list(x.shape) == rec["shape"] and (scale, pad_left, pad_top) == (rec["scale"], rec["pad_left"], rec["pad_top"])
    assert hashlib.sha256((x[0] * 255.0).round().to(torch.uint8).numpy().tobytes()).hexdigest() == rec["u8_sha256"]
    assert float(x.double().sum()) == rec["sum"]
    if name == "word.png":
        tw, th, sc, rw, rh, pl, pt = D.letterbox_geometry(1919, 1079, 640)
        assert (rw, rh) == (640, 359) and rh != round(1079 * sc)       # the truncation the odd size is there for


@pytest.mark.parametrize("name", sorted(GOLD["images"]))
def test_oracle_predict_equals_reference_predict_on_its_own_images(name, standin):
    """oracle.detector_ref.predict (restatement of ref:util/yolov9.py:115-136) == the reference's YOLOv9Detector.predict on the same
    image and the same stand-in blob: same boxes, same scores, BIT FOR BIT (when the blob regenerated here is the generator's)."""
    from PIL import Image
    from oracle import detector_ref as D
    model, same_blob = standin
    rec = GOLD["images"][name]["predict"]
    b, s, c = D.predict(model, Image.open(IMG_DIR / name), conf=GOLD["conf"], imgsz=GOLD["imgsz"], iou=GOLD["nms_iou"])
    ref_b = torch.tensor(rec["boxes_bits"], dtype=torch.int32).view(torch.float32).reshape(-1, 4)
    ref_s = torch.tensor(rec["conf_bits"], dtype=torch.int32).view(torch.float32)
    if not same_blob:
        pytest.skip("the stand-in blob regenerated on this machine differs from the generator's (other torch build): bitwise vectors do not apply")
    assert b.shape == ref_b.shape and torch.equal(b, ref_b) and torch.equal(s, ref_s)


class _Det:
    def __init__(self, b, s): self.b, self.s = b, s
    def predict(self, source, conf, iou, imgsz=None):
        return [types.SimpleNamespace(boxes=types.SimpleNamespace(xyxy=self.b, conf=self.s))]


class _RecCap:
    config = types.SimpleNamespace(name_or_path="florence-recorder", model_type="florence2")
    device = torch.device("cpu")

    def __init__(self): self.boxes = []

    def caption_crops(self, image, boxes, max_new_tokens=20, batch_size=128):
        self.boxes = [list(b) for b in boxes]
        return torch.cat([torch.arange(len(boxes[s:s + batch_size])) for s in range(0, len(boxes), batch_size)]).view(-1, 1)


class _Proc:
    def batch_decode(self, ids, skip_special_tokens=True): return [f" cap{int(i)} " for i in ids.view(-1)]


@pytest.mark.parametrize("name", sorted(GOLD["images"]))
def test_host_glue_equals_reference_get_som_labeled_img_on_its_own_images(name):
    """omniparser_amd.util.utils.get_som_labeled_img (host twin of ref:util/utils.py:417-496) fed the reference's final boxes: same
    elements in the same order, same label keys, and the crop rectangles it hands the captioner have the shapes the reference's
    cv2.resize calls saw (ref:util/utils.py:95-102, f32 truncation)."""
    from PIL import Image
    from omniparser_amd.synth import synthetic_ocr
    from omniparser_amd.util.utils import get_som_labeled_img
    g = GOLD["images"][name]
    b = torch.tensor(g["predict"]["boxes_bits"], dtype=torch.int32).view(torch.float32).reshape(-1, 4)
    s = torch.tensor(g["predict"]["conf_bits"], dtype=torch.int32).view(torch.float32)
    img = Image.open(IMG_DIR / name)
    texts, obox = synthetic_ocr(GOLD["ocr"]["seed"], img.size[0], img.size[1], GOLD["ocr"]["n"])
    cap = _RecCap()
    enc, lab, elems = get_som_labeled_img(img, _Det(b, s), BOX_TRESHOLD=GOLD["conf"], output_coord_in_ratio=True, ocr_bbox=obox,
                                          caption_model_processor={"model": cap, "processor": _Proc()}, ocr_text=texts,
                                          use_local_semantics=True, iou_threshold=0.7, scale_img=False, batch_size=128)
    assert elems == g["som"]["elems"]
    assert list(lab.keys()) == g["som"]["label_keys"]
    shapes = [[y1 - y0, x1 - x0] for (x0, y0, x1, y1) in cap.boxes]
    assert shapes == g["som"]["crop_shapes_hw"] and len(shapes) == g["som"]["n_crops"]
