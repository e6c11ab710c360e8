"""Seeded synthetic inputs (no datasets on the box): GUI-like 1920x1080 screenshots and an OCR-box
fixture (the reference crashes with empty OCR — ref:util/utils.py:437-444 — and OCR itself is out of
scope)."""
import numpy as np

# Seeds of the 8 screenshots of the bench batch (bench.py, tests/gpu_checks.py::check_bench_path).  Chosen by tools/scan_parity_frames.py
# (CPU oracle only, criteria fixed beforehand, profiles/r5_parity_frame_scan.md) among seeds 8..109 — the stand-in's calibration batch holds
# seeds 0..7, so every benched frame is HELD OUT — on which the oracle's own NMS takes NO decision on a tie (no IoU within 1e-5 of the
# threshold, no suppression by a box whose score is within 4e-6 of its victim's) and whose margins (threshold >= 1e-4, IoU >= 3e-5, score
# gap >= 9e-6) are several times the GPU-vs-oracle differences (scores 1e-6, boxes 2e-4 px, logits 2e-5): "box for box against the oracle's
# own list" is a well-posed statement on every frame of the benched batch.  29 of the 110 scanned seeds pass (round 4's stand-in: 9); of
# the 26 held-out ones, the 8 whose crop counts sum to 345 per batch — the crop count rounds 3-4 benched (128 + 128 + 96-row plans), so
# screenshots/s stays comparable across rounds.
BENCH_SEEDS = (14, 15, 18, 51, 87, 95, 102, 104)


def synthetic_screenshot(seed: int = 0, w: int = 1920, h: int = 1080) -> np.ndarray:
    """uint8 [h, w, 3] RGB: flat panels + 150 filled icon-like rectangles + 40 text-like noise strips."""
    rng = np.random.default_rng(seed)
    img = np.empty((h, w, 3), dtype=np.uint8)
    img[:] = rng.integers(200, 256, size=3, dtype=np.uint8)
    # a few large panels
    for _ in range(6):
        x0, y0 = int(rng.integers(0, w - 200)), int(rng.integers(0, h - 100))
        pw, ph = int(rng.integers(200, w // 2)), int(rng.integers(60, h // 2))
        img[y0:y0 + ph, x0:x0 + pw] = rng.integers(120, 256, size=3, dtype=np.uint8)
    for _ in range(150):
        s = int(rng.integers(16, 65))
        x0, y0 = int(rng.integers(0, w - s)), int(rng.integers(0, h - s))
        col = rng.integers(0, 256, size=3, dtype=np.uint8)
        img[y0:y0 + s, x0:x0 + s] = col
        if s >= 24:  # inner glyph
            q = s // 4
            img[y0 + q:y0 + s - q, x0 + q:x0 + s - q] = 255 - col
    for _ in range(40):
        tw, th = int(rng.integers(60, 300)), int(rng.integers(10, 22))
        x0, y0 = int(rng.integers(0, w - tw)), int(rng.integers(0, h - th))
        strip = rng.integers(0, 2, size=(th, tw, 1), dtype=np.uint8) * rng.integers(100, 256, dtype=np.uint8)
        img[y0:y0 + th, x0:x0 + tw] = np.broadcast_to(255 - strip, (th, tw, 3))
    return img


def synthetic_ocr(seed: int = 0, w: int = 1920, h: int = 1080, n: int = 40):
    """(texts, xyxy integer pixel boxes) on a jittered grid — non-overlapping, like EasyOCR output."""
    rng = np.random.default_rng(1000 + seed)
    cols, rows = 8, (n + 7) // 8
    cw, rh = w // cols, h // rows
    texts, boxes = [], []
    for i in range(n):
        cx, cy = (i % cols) * cw, (i // cols) * rh
        bw, bh = int(rng.integers(cw // 4, cw // 2)), int(rng.integers(12, 28))
        x0 = cx + int(rng.integers(0, cw - bw))
        y0 = cy + int(rng.integers(0, rh - bh))
        boxes.append([x0, y0, x0 + bw, y0 + bh])
        texts.append(f"t{i}")
    return texts, boxes
