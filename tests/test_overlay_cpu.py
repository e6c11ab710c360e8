"""Set-of-marks overlay: the product's draw list must equal the cv2 call sequence of the
REFERENCE's own util/box_annotator.py, recorded by tests/golden/gen_overlay_golden.py under a cv2 stub."""
import json
from pathlib import Path

import numpy as np
import torch

from omniparser_amd.util import overlay as OV

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_overlay.json").read_text())["cases"]


def _cmds(case):
    xyxy = np.array(case["xyxy"], dtype=np.float32).reshape(-1, 4)
    return OV.plan_overlay(xyxy, case["labels"], (case["w"], case["h"]), avoid_overlap=case["avoid_overlap"], **case["cfg"])


def test_draw_list_equals_reference_call_sequence():
    assert len(GOLD) >= 30
    seen_positions = set()
    for case in GOLD:
        got = json.loads(json.dumps(_cmds(case)))
        assert got == case["calls"], (case["w"], case["h"], case["cfg"], case["avoid_overlap"])
        for c in got:
            if c[0] == "rect" and c[4] == OV.FILLED:
                seen_positions.add((c[1][0] < c[2][0], c[1][1] < c[2][1]))
    assert sum(len(c["calls"]) for c in GOLD) > 3000


def test_all_four_placements_and_fallback_are_exercised():
    hist = np.zeros(4, dtype=int)
    fallback = 0
    for case in GOLD:
        if not case["avoid_overlap"] or not case["xyxy"] or len(case["labels"]) != len(case["xyxy"]):
            continue
        b = np.array(case["xyxy"], dtype=np.float32).astype(int)
        cfg = case["cfg"]
        sizes = np.array([OV.hershey_text_size(t, cfg["text_scale"], cfg["text_thickness"]) for t in case["labels"]])
        cand = OV.label_candidates(b, sizes, cfg["text_padding"])
        pick = OV.choose_labels(b, sizes, cfg["text_padding"], (case["w"], case["h"]))
        tags = cand[:, :, 2:].reshape(-1, 4)
        bad = ((OV._max_overlap_ratio(tags, b) > 0.3).any(1) | (tags[:, 0] < 0) | (tags[:, 2] > case["w"]) | (tags[:, 1] < 0)
               | (tags[:, 3] > case["h"])).reshape(-1, 4)
        fallback += int(bad.all(1).sum())
        for i in range(len(b)):
            hist[[j for j in range(4) if (cand[i, j] == pick[i]).all()][0]] += 1
    assert (hist > 50).all() and fallback > 10, (hist, fallback)


def test_text_size_formula_and_palette():
    assert OV.hershey_text_size("7", 1.0, 1) == (21, 22)          # 20*1+1 ; 21 + 1 = 22
    assert OV.hershey_text_size("123", 0.5, 2) == (32, 12)         # 30+2 ; 10.5+1.5 = 12
    assert OV.hershey_text_size("12", 0.48, 1) == (20, 11)         # rint(20.2) ; rint(11.08)
    assert OV.PALETTE_RGB.shape == (21, 3) and tuple(OV.PALETTE_RGB[0]) == (0xA3, 0x51, 0xFB)


def test_render_touches_only_planned_pixels_and_annotate_contract():
    from omniparser_amd.util.utils import annotate
    img = np.full((200, 320, 3), 50, dtype=np.uint8)
    boxes = torch.tensor([[0.25, 0.5, 0.2, 0.3], [0.7, 0.4, 0.1, 0.2]])          # cxcywh ratios
    frame, coords = annotate(img, boxes, None, ["a", "b"], text_scale=0.4, text_padding=5)
    assert frame.shape == img.shape and frame.dtype == np.uint8 and (img == 50).all()       # input untouched
    assert list(coords) == ["a", "b"]
    np.testing.assert_allclose(coords["a"], [0.15 * 320, 0.35 * 200, 0.2 * 320, 0.3 * 200], rtol=1e-6)
    changed = (frame != 50).any(-1)
    assert changed.any()
    ys, xs = np.nonzero(changed)
    # everything drawn lies within the union of outlines (+stroke) and tags of the plan
    cmds = OV.plan_overlay(np.array([[48, 70, 112, 130], [208, 60, 240, 100]], dtype=np.float32), ["0", "1"], (320, 200),
                           text_scale=0.4, text_padding=5, text_thickness=2, thickness=3)
    allowed = np.zeros_like(changed)
    for c in cmds:
        if c[0] == "rect":
            (x1, y1), (x2, y2) = c[1], c[2]
            allowed[max(y1 - 2, 0):y2 + 3, max(x1 - 2, 0):x2 + 3] = True
    assert allowed[ys, xs].all()
    outline = tuple(int(v) for v in OV.PALETTE_RGB[0][::-1])      # palette colour handed over in B,G,R order (ref quirk)
    assert tuple(frame[100, 48]) == outline


def test_empty_and_degenerate_inputs():
    assert OV.plan_overlay(np.zeros((0, 4), np.float32), [], (100, 100)) == []
    scene = np.zeros((50, 60, 3), np.uint8)
    assert OV.BoxAnnotator().annotate(scene, np.zeros((0, 4), np.float32), labels=[]).sum() == 0
    cmds = OV.plan_overlay(np.array([[10.9, 10.2, 10.1, 30.7]], np.float32), ["0"], (60, 50))     # zero width after truncation
    assert cmds[0][1:3] == ((10, 10), (10, 30))
    OV.render(scene, cmds)
