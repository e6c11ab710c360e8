"""`-m gpu`: the HIP path on the REFERENCE'S OWN images (tests/golden/ref_imgs: word.png 1919x1079 RGBA, demo_image.jpg 3240x2160 =
BASELINE configs[0]) behind the reference API, against the oracle pipeline on the same images — the oracle itself is pinned to the
reference's own output on these images by tests/test_reference_images_cpu.py (tests/golden/reference_images.json)."""
import json
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent
GOLD = json.loads((HERE / "golden" / "reference_images.json").read_text())


@pytest.mark.parametrize("name", ["word.png", "demo_image.jpg"])
def test_reference_image_end_to_end_vs_oracle(name):
    """get_som_labeled_img (full-width YOLOv9-E stand-in, 768x768 caption crops = the reference's CPU path) on the image: the same
    elements as the CPU pipeline, IoU >= 0.999, identical types / sources / order pairing, and greedy captions token-exact on the
    crops the CPU captioner's budget covers (8 crops, 3.6 s each)."""
    from PIL import Image
    import gpu_checks as G
    from omniparser_amd.synth import synthetic_ocr
    img = Image.open(HERE / "golden" / "ref_imgs" / name)
    assert list(img.size) == GOLD["images"][name]["size"]
    ocr = synthetic_ocr(GOLD["ocr"]["seed"], img.size[0], img.size[1], GOLD["ocr"]["n"])
    out = G.check_end_to_end(width=1.0, R=768, image=img, ocr=ocr, max_crops_checked=8)
    print(name, out)
    assert out["n_gpu"] == out["n_ref"] and out["min_iou"] >= 0.999, out
    assert out["captioned"] >= 6 and out["identical_crops_token_exact"] >= out["captioned"] - 1, out
    # the CPU pipeline here is the one whose detector output equals the reference's on this image (same count as the golden vector)
    assert out["boxes_ref"] == GOLD["images"][name]["predict"]["n"], out
