"""Auto-activating pins for the two third-party functions the oracle RESTATES because their packages are absent from this image
(VERDICT r5 "missing" item 3):

  * `cv2.resize(crop, (64, 64))` (default INTER_LINEAR, 8-bit)  — ref:util/utils.py:102  -> oracle/preprocess_ref.py::cv2_resize_linear
  * `torchvision.ops.batched_nms` / `nms`                       — ref:util/yolov9.py:9,131 -> oracle/detector_ref.py::batched_nms / nms

Each test is `pytest.importorskip`-gated: on this image both skip (neither package exists, there is no network); the day a box has
opencv-python or torchvision installed they run by themselves and hold the restatements to the real library bit for bit.  Until then
DESIGN.md section 4 keeps the words "parity unpinned" for these two functions."""
import numpy as np
import pytest
import torch


def _crops(rng):
    """crop shapes the hand-off produces (h, w): tiny, ragged, the exact-2x case OpenCV routes to INTER_AREA (128x128 -> 64x64, and
    128 on one axis only, which stays on the linear path), up- and down-sampling on either axis, a screenshot-sized crop."""
    shapes = [(1, 1), (1, 9), (7, 1), (2, 2), (5, 7), (31, 33), (63, 65), (64, 64), (128, 128), (128, 64), (64, 128), (128, 200),
              (127, 129), (256, 256), (200, 37), (37, 200), (480, 640), (1080, 1920), (13, 700)]
    for h, w in shapes:
        yield rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        g = np.zeros((h, w, 3), dtype=np.uint8)                     # smooth content (what a GUI crop looks like) + a hard edge
        g[..., 0] = (np.arange(w)[None, :] * 255 // max(w - 1, 1)).astype(np.uint8)
        g[..., 1] = (np.arange(h)[:, None] * 255 // max(h - 1, 1)).astype(np.uint8)
        g[h // 2:, w // 2:, 2] = 255
        yield g


def test_cv2_resize_linear_restatement_is_bit_exact_against_opencv():
    cv2 = pytest.importorskip("cv2")
    from oracle import preprocess_ref as PR
    rng = np.random.default_rng(0)
    worst = 0
    for crop in _crops(rng):
        want = cv2.resize(crop, (64, 64))                           # the reference's literal call (ref:util/utils.py:102)
        got = PR.cv2_resize_linear(crop, (64, 64))
        worst = max(worst, int(np.abs(want.astype(np.int32) - got.astype(np.int32)).max()))
        assert np.array_equal(want, got), f"crop {crop.shape}: max abs diff {worst}"
    # and through the whole crop pre-processing of one synthetic frame (what the caption oracle is keyed on)
    from omniparser_amd.synth import synthetic_screenshot
    img = synthetic_screenshot(0, 1920, 1080)
    for box in [(0, 0, 1920, 1080), (100, 50, 228, 178), (17, 900, 400, 1079), (1900, 0, 1920, 20)]:
        x0, y0, x1, y1 = box
        assert np.array_equal(cv2.resize(img[y0:y1, x0:x1, :], (64, 64)), PR.cv2_resize_linear(img[y0:y1, x0:x1, :], (64, 64))), box


def test_cv2_resize_empty_crop_raises_like_opencv():
    cv2 = pytest.importorskip("cv2")
    from oracle import preprocess_ref as PR
    empty = np.zeros((0, 5, 3), dtype=np.uint8)
    with pytest.raises(cv2.error):
        cv2.resize(empty, (64, 64))
    with pytest.raises(ValueError):
        PR.cv2_resize_linear(empty, (64, 64))


def _clouds(rng):
    """(boxes, scores, class ids, iou): empty, one box, duplicates (score ties + IoU = 1), touching boxes (IoU = 0), a dense cloud below
    torchvision's 4000-numel switch (coordinate-offset trick) and one above it (per-class loop), several classes, degenerate boxes."""
    def cloud(n, nc, extent=640.0, size=60.0):
        c = rng.uniform(0, extent, size=(n, 2)).astype(np.float32)
        wh = rng.uniform(1, size, size=(n, 2)).astype(np.float32)
        b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        s = rng.uniform(0.05, 1.0, size=n).astype(np.float32)
        k = rng.integers(0, nc, size=n)
        return torch.from_numpy(b), torch.from_numpy(s), torch.from_numpy(k)
    yield torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.int64), 0.1
    yield torch.tensor([[0., 0., 10., 10.]]), torch.tensor([0.5]), torch.tensor([0]), 0.1
    b = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 10.], [10., 0., 20., 10.], [5., 5., 5., 5.], [0., 0., 10., 10.]])
    yield b, torch.tensor([0.9, 0.9, 0.9, 0.3, 0.2]), torch.tensor([0, 0, 0, 0, 1]), 0.1
    for n, nc, iou in ((50, 1, 0.1), (999, 1, 0.1), (1000, 3, 0.7), (1001, 1, 0.1), (1400, 1, 0.1), (3000, 4, 0.45), (9000, 1, 0.1)):
        b, s, k = cloud(n, nc)
        s[::7] = s[0]                                               # exact score ties: the stable sort decides
        yield b, s, k, iou


def test_batched_nms_restatement_equals_torchvision():
    tv = pytest.importorskip("torchvision")
    from oracle import detector_ref as D
    rng = np.random.default_rng(1)
    for boxes, scores, idxs, iou in _clouds(rng):
        want = tv.ops.batched_nms(boxes, scores, idxs, iou)
        got = D.batched_nms(boxes, scores, idxs, iou)
        assert torch.equal(want, got), f"{boxes.shape[0]} boxes, {int(idxs.max()) + 1 if idxs.numel() else 0} classes, iou {iou}: {want.numel()} vs {got.numel()} keeps"
        if boxes.shape[0]:
            assert torch.equal(tv.ops.nms(boxes, scores, iou), D.nms(boxes, scores, iou))


def test_pins_are_inactive_only_because_the_packages_are_absent():
    """the two tests above must not skip for any other reason: when a package IS importable its test runs (no silent gate)."""
    import importlib.util
    state = {name: importlib.util.find_spec(name) is not None for name in ("cv2", "torchvision")}
    # nothing to assert about presence; recorded so that a CI log shows which pins were live
    print("third-party pins live:", state)
    from oracle import preprocess_ref as PR
    from oracle import detector_ref as D
    assert callable(PR.cv2_resize_linear) and callable(D.batched_nms)
