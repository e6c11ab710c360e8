"""ORACLE (test infrastructure): CPU restatement of the crop pre-processing between the two models.

ref:util/utils.py:97-105   int(x*W) truncation, HWC slice, cv2.resize(crop, (64, 64)) (INTER_LINEAR)
ref:util/utils.py:120-123  processor(images=..., [do_resize=False]) -> CLIP image processor:
                           CPU branch resizes 64x64 -> 768x768 BICUBIC (Pillow), rescale 1/255, normalise.

cv2 (opencv-python, unpinned in ref:requirements.txt:11-12) is absent here: `cv2_resize_linear` restates
OpenCV's generic 8-bit INTER_LINEAR path (imgproc/resize.cpp: 11-bit fixed-point coefficients,
horizontal pass in int32, vertical `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`).  PARITY UNPINNED
for that function (no cv2 to compare with; IPP-enabled builds may differ in the last bit); the Pillow and
transformers steps are the real libraries.  tests/test_third_party_pins_cpu.py holds the pin that activates by itself where
opencv-python is importable (`pytest.importorskip("cv2")`: 19 crop shapes incl. the exact-2x case OpenCV routes to INTER_AREA).
"""
import numpy as np
from PIL import Image


def _coeffs(src, dst):
    scale = src / dst
    idx = np.zeros(dst, dtype=np.int64)
    frac = np.zeros(dst, dtype=np.float32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - s)
        idx[d], frac[d] = s, f
    return idx, frac


def _short(v):
    return np.clip(np.rint(v), -32768, 32767).astype(np.int64)   # saturate_cast<short>: round half to even


def cv2_resize_linear(src: np.ndarray, dsize=(64, 64)) -> np.ndarray:
    """src uint8 [h,w,3] -> uint8 [dh,dw,3]."""
    dw, dh = dsize
    h, w = src.shape[:2]
    if h == 0 or w == 0:
        raise ValueError("empty crop (cv2.resize raises; the reference swallows it, ref:util/utils.py:104-105)")
    sx, fx = _coeffs(w, dw)
    # horizontal: fx zeroed at the borders
    lo = sx < 0
    fx[lo] = 0; sx[lo] = 0
    hi = sx >= w - 1
    fx[hi] = 0; sx[hi] = w - 1
    a0 = _short((np.float32(1) - fx) * np.float32(2048)); a1 = _short(fx * np.float32(2048))
    sx1 = np.minimum(sx + 1, w - 1)
    s = src.astype(np.int64)
    rows = s[:, sx] * a0[None, :, None] + s[:, sx1] * a1[None, :, None]          # [h, dw, 3] int
    # vertical: beta from the un-clamped fraction, row indices clipped
    sy, fy = _coeffs(h, dh)
    b0 = _short((np.float32(1) - fy) * np.float32(2048)); b1 = _short(fy * np.float32(2048))
    y0 = np.clip(sy, 0, h - 1); y1 = np.clip(sy + 1, 0, h - 1)
    S0, S1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_box_pixels(coord, W, H):
    """ref:util/utils.py:99-100: ratio xyxy -> integer pixel box by truncation."""
    return int(coord[0] * W), int(coord[1] * H), int(coord[2] * W), int(coord[3] * H)


def caption_pixel_values(image: np.ndarray, box_px, R: int, mean, std) -> np.ndarray:
    """[R,R,3] float32 exactly as the reference feeds Florence-2 (HWC here; the model gets CHW)."""
    x0, y0, x1, y1 = box_px
    crop = image[y0:y1, x0:x1, :]
    c64 = cv2_resize_linear(crop, (64, 64))
    if R != 64:
        c64 = np.asarray(Image.fromarray(c64).resize((R, R), Image.Resampling.BICUBIC))
    x = (c64.astype(np.float64) * (1 / 255)).astype(np.float32)        # hf image_transforms.rescale
    return ((x - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)).astype(np.float32)
