"""Set-of-marks overlay: box outlines + numbered label tags on the screenshot.

Behavioural restatement of ref:util/box_annotator.py:10-262 (+ the call site ref:util/utils.py:336-364), split
the MI355X-service way: a pure, vectorised *layout* step (`plan_overlay`, numpy, O(K²) integer overlap tests for
all K labels x 4 candidate positions at once) that yields a draw list, and a *raster* step (`render`, Pillow)
that executes it.  The draw list is the parity surface: tests/golden/reference_overlay.json holds the
reference's own cv2 call sequence (its source executed under a recording cv2 stub,
tests/golden/gen_overlay_golden.py) and `plan_overlay` must reproduce it call for call.

cv2 and supervision are absent in this image, so three third-party pieces are restated from memory and are
UNPINNED (stated here and in DESIGN.md): OpenCV's `getTextSize` for FONT_HERSHEY_SIMPLEX (digits only — the
labels are box indices), supervision 0.18's default colour palette, and cv2's stroke rasterisation (Pillow
strokes are used; pixel output is a visual aid, the draw list is what is compared).
"""
from typing import List, Sequence, Tuple

import numpy as np
from PIL import Image, ImageDraw, ImageFont

# supervision==0.18.0 `ColorPalette.DEFAULT` (hex, RGB) — restated from memory, unpinned.
_PALETTE_HEX = ("A351FB", "FF4040", "FFA1A0", "FF7633", "FFB633", "D1D435", "4CFB12", "94CF1A", "40DE8A", "1B9640", "00D6C1",
                "2E9CAA", "00C4FF", "364797", "6675FF", "0019EF", "863AFF", "530087", "CD3AFF", "FF97CA", "FF39C9")
PALETTE_RGB = np.array([[int(h[i:i + 2], 16) for i in (0, 2, 4)] for h in _PALETTE_HEX], dtype=np.int64)

RECT, TEXT = "rect", "text"
FILLED = -1
LABEL_OVERLAP_LIMIT = 0.3          # ref:util/box_annotator.py:191


def _round_half_even(v: float) -> int:
    return int(np.rint(v))         # cvRound == lrint under the default rounding mode


def hershey_text_size(text: str, scale: float, thickness: int) -> Tuple[int, int]:
    """cv2.getTextSize(text, FONT_HERSHEY_SIMPLEX, scale, thickness)[0] for digit strings: every digit of the
    simplex face advances 20 units, cap line 12 + base line 9 (OpenCV hershey table; from memory, unpinned)."""
    width = _round_half_even(20.0 * len(text) * scale + thickness)
    height = _round_half_even((12 + 9) * scale + (thickness + 1) / 2)
    return width, height


def _max_overlap_ratio(tags: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    """ref:util/box_annotator.py:166-178 `IoU(..., return_max=True)` for every tag [T,4] against every box [K,4]
    -> [T,K] float64: max(inter/union, inter/area_tag, inter/area_box), the two ratios only when both areas > 0."""
    t = tags[:, None, :].astype(np.int64)
    b = boxes[None, :, :].astype(np.int64)
    iw = np.maximum(0, np.minimum(t[..., 2], b[..., 2]) - np.maximum(t[..., 0], b[..., 0]))
    ih = np.maximum(0, np.minimum(t[..., 3], b[..., 3]) - np.maximum(t[..., 1], b[..., 1]))
    inter = (iw * ih).astype(np.float64)
    at = ((t[..., 2] - t[..., 0]) * (t[..., 3] - t[..., 1])).astype(np.float64)
    ab = ((b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])).astype(np.float64)
    union = at + ab - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / union
        both = (at > 0) & (ab > 0)
        r1 = np.where(both, inter / at, 0.0)
        r2 = np.where(both, inter / ab, 0.0)
    return np.maximum(iou, np.maximum(r1, r2))


def label_candidates(xyxy: np.ndarray, sizes: np.ndarray, pad: int) -> np.ndarray:
    """All four tag placements in the reference's trial order (top-left above the box, outer-left, outer-right,
    top-right above the box; ref:util/box_annotator.py:203-259) -> int64 [K,4,6] =
    (text_x, text_y, tag_x1, tag_y1, tag_x2, tag_y2)."""
    x1, y1, x2 = xyxy[:, 0], xyxy[:, 1], xyxy[:, 2]
    tw, th = sizes[:, 0], sizes[:, 1]
    above, below = y1 - 2 * pad - th, y1 + 2 * pad + th
    cand = np.stack([
        np.stack([x1 + pad, y1 - pad, x1, above, x1 + 2 * pad + tw, y1], -1),
        np.stack([x1 - pad - tw, y1 + pad + th, x1 - 2 * pad - tw, y1, x1, below], -1),
        np.stack([x2 + pad, y1 + pad + th, x2, y1, x2 + 2 * pad + tw, below], -1),
        np.stack([x2 - pad - tw, y1 - pad, x2 - 2 * pad - tw, above, x2, y1], -1),
    ], 1)
    return cand.astype(np.int64)


def choose_labels(xyxy: np.ndarray, sizes: np.ndarray, pad: int, image_size: Tuple[int, int]) -> np.ndarray:
    """First placement whose tag neither covers any detection box by more than 0.3 (max-ratio IoU) nor leaves the
    image; the last one when all four fail (ref:util/box_annotator.py:181-262).  -> int64 [K,6]."""
    K = xyxy.shape[0]
    if K == 0:
        return np.zeros((0, 6), dtype=np.int64)
    cand = label_candidates(xyxy, sizes, pad)
    tags = cand[:, :, 2:].reshape(K * 4, 4)
    hit = (_max_overlap_ratio(tags, xyxy) > LABEL_OVERLAP_LIMIT).any(1)
    W, H = image_size
    outside = (tags[:, 0] < 0) | (tags[:, 2] > W) | (tags[:, 1] < 0) | (tags[:, 3] > H)
    bad = (hit | outside).reshape(K, 4)
    pick = np.where(bad.all(1), 3, np.argmin(bad, 1))
    return cand[np.arange(K), pick]


def plan_overlay(xyxy: np.ndarray, labels: Sequence[str], image_size: Tuple[int, int], text_scale=0.5, text_padding=10,
                 text_thickness=2, thickness=3, avoid_overlap=True, text_size=hershey_text_size) -> List[tuple]:
    """Draw list of ref:util/box_annotator.py:86-162 in call order: per detection one outline
    (RECT, (x1,y1), (x2,y2), colour, thickness), one filled tag (RECT, ..., FILLED) and one
    (TEXT, label, (x,y) baseline origin, colour, scale, thickness).  Colours are the tuples the reference hands to
    cv2: the palette entry in B,G,R order (drawn as-is on the RGB frame — reference quirk kept) and a black/white
    text colour from the RGB luminance (> 160 -> black)."""
    boxes = np.asarray(xyxy).reshape(-1, 4).astype(int)              # truncation toward zero, as `.astype(int)`
    K = boxes.shape[0]
    texts = [str(labels[i]) if labels is not None and len(labels) == K else "None" for i in range(K)]
    sizes = np.array([text_size(t, text_scale, text_thickness) for t in texts], dtype=np.int64).reshape(K, 2)
    if avoid_overlap:
        spots = choose_labels(boxes, sizes, text_padding, image_size)
    else:
        spots = label_candidates(boxes, sizes, text_padding)[:, 0] if K else np.zeros((0, 6), dtype=np.int64)
    rgb = PALETTE_RGB[np.arange(K) % len(PALETTE_RGB)]
    lum = 0.299 * rgb[:, 0] + 0.587 * rgb[:, 1] + 0.114 * rgb[:, 2]
    cmds = []
    for i in range(K):
        bgr = (int(rgb[i, 2]), int(rgb[i, 1]), int(rgb[i, 0]))
        x1, y1, x2, y2 = (int(v) for v in boxes[i])
        tx, ty, a1, b1, a2, b2 = (int(v) for v in spots[i])
        cmds.append((RECT, (x1, y1), (x2, y2), bgr, int(thickness)))
        cmds.append((RECT, (a1, b1), (a2, b2), bgr, FILLED))
        cmds.append((TEXT, texts[i], (tx, ty), (0, 0, 0) if lum[i] > 160 else (255, 255, 255), float(text_scale), int(text_thickness)))
    return cmds


_FONT_CACHE = {}
_MASK_CACHE = {}
PRIM_FILL, PRIM_RING, PRIM_MASK = 0, 1, 2          # csrc/overlay_png.hip


def _digit_font(scale: float):
    """Scalable Pillow face whose cap height matches the simplex face (12 units * scale above the baseline... the
    Hershey digits span 21 units; Pillow's built-in Aileron cap height is ~0.72 em)."""
    px = max(int(round(21 * scale / 0.72)), 6)
    if px not in _FONT_CACHE:
        try:
            _FONT_CACHE[px] = ImageFont.load_default(size=px)
        except TypeError:                                            # Pillow < 10.1: bitmap face only
            _FONT_CACHE[px] = ImageFont.load_default()
    return _FONT_CACHE[px]


def text_mask(text: str, scale: float):
    """8-bit coverage mask of `text` as Pillow's ImageDraw.text(..., anchor="ls") rasters it, and its offset from the text origin
    (left end of the baseline): drawn once with ink 255 on a black canvas — Pillow blends out = round(in (255 - m) / 255 + ink m / 255),
    which returns m itself there.  -> (uint8 [h,w], dx, dy); an empty mask for whitespace."""
    key = (text, float(scale))
    if key not in _MASK_CACHE:
        font = _digit_font(scale)
        px = getattr(font, "size", 12)
        ox, oy = 2 * px, 3 * px
        canvas = Image.new("L", (ox + px * (len(text) + 2), oy + 2 * px), 0)
        draw = ImageDraw.Draw(canvas)
        try:
            draw.text((ox, oy), text, fill=255, font=font, anchor="ls")       # cv2 origin = left end of the baseline
        except (ValueError, TypeError):
            draw.text((ox, oy - 10), text, fill=255, font=font)
        box = canvas.getbbox()
        if box is None:
            _MASK_CACHE[key] = (np.zeros((0, 0), dtype=np.uint8), 0, 0)
        else:
            _MASK_CACHE[key] = (np.ascontiguousarray(np.asarray(canvas.crop(box))), box[0] - ox, box[1] - oy)
    return _MASK_CACHE[key]


def raster_primitives(cmds: Sequence[tuple]):
    """Draw list -> the primitive list both rasters execute IN ORDER (host: `render`; device: OMNI_OP_OVERLAY):
    int32 [n,8] = {kind, x0, y0, x1, y1, r | g << 8 | b << 16, a, 0} + the concatenated coverage masks (uint8).
      FILL  pixels x0..x1, y0..y1 (inclusive) take the colour;
      RING  an outline of width a: the pixels of the rectangle that are NOT in its interior shrunk by a on every side — cv2 centres a
            stroke of thickness t on the edge, hence the rectangle grown by t // 2 (what ImageDraw.rectangle(outline, width) draws
            for every rectangle at least 2 a wide and high; smaller ones are simply filled);
      MASK  x1, y1 exclusive, a = offset of its (y1-y0) x (x1-x0) mask: out = (t + (t >> 8)) >> 8, t = in (255 - m) + ink m + 128
            (Pillow's BLEND8).
    The colour tuple of a command is written to the frame's channels as given (the reference hands B,G,R tuples to an RGB frame)."""
    prims, masks, off = [], [], 0
    for c in cmds:
        if c[0] == RECT:
            _, (x1, y1), (x2, y2), col, t = c
            if x2 < x1:
                x1, x2 = x2, x1
            if y2 < y1:
                y1, y2 = y2, y1
            ink = int(col[0]) | int(col[1]) << 8 | int(col[2]) << 16
            if t == FILLED:
                prims.append((PRIM_FILL, x1, y1, x2, y2, ink, 0, 0))
            else:
                o = t // 2
                prims.append((PRIM_RING, x1 - o, y1 - o, x2 + o, y2 + o, ink, max(t, 1), 0))
        else:
            _, text, (x, y), col, scale, _t = c
            m, dx, dy = text_mask(text, scale)
            if m.size == 0:
                continue
            ink = int(col[0]) | int(col[1]) << 8 | int(col[2]) << 16
            prims.append((PRIM_MASK, x + dx, y + dy, x + dx + m.shape[1], y + dy + m.shape[0], ink, off, 0))
            masks.append(m.reshape(-1))
            off += m.size
    P = np.asarray(prims, dtype=np.int32).reshape(-1, 8)
    M = np.concatenate(masks) if masks else np.zeros(1, dtype=np.uint8)
    return P, M


def render(scene: np.ndarray, cmds: Sequence[tuple]) -> np.ndarray:
    """Raster the draw list onto `scene` (uint8 [H,W,3], modified in place like cv2 does) on the host: the primitive semantics of
    `raster_primitives`, executed with numpy.  The device raster (`render_device`) produces the same bytes."""
    H, W = scene.shape[:2]
    P, M = raster_primitives(cmds)
    for kind, x0, y0, x1, y1, ink, a, _ in P.tolist():
        col = np.array([ink & 255, (ink >> 8) & 255, (ink >> 16) & 255], dtype=np.uint8)
        if kind == PRIM_MASK:
            cx0, cy0, cx1, cy1 = max(x0, 0), max(y0, 0), min(x1, W), min(y1, H)
            if cx0 >= cx1 or cy0 >= cy1:
                continue
            m = M[a:a + (y1 - y0) * (x1 - x0)].reshape(y1 - y0, x1 - x0)[cy0 - y0:cy1 - y0, cx0 - x0:cx1 - x0].astype(np.uint32)[..., None]
            px = scene[cy0:cy1, cx0:cx1].astype(np.uint32)
            t = px * (255 - m) + col.astype(np.uint32) * m + 128
            scene[cy0:cy1, cx0:cx1] = np.where(m > 0, (t + (t >> 8)) >> 8, px).astype(np.uint8)
            continue
        cx0, cy0, cx1, cy1 = max(x0, 0), max(y0, 0), min(x1, W - 1), min(y1, H - 1)
        if cx0 > cx1 or cy0 > cy1:
            continue
        if kind == PRIM_FILL or x0 + a > x1 - a or y0 + a > y1 - a:
            scene[cy0:cy1 + 1, cx0:cx1 + 1] = col
            continue
        for (bx0, by0, bx1, by1) in ((x0, y0, x1, y0 + a - 1), (x0, y1 - a + 1, x1, y1), (x0, y0, x0 + a - 1, y1), (x1 - a + 1, y0, x1, y1)):
            bx0, by0, bx1, by1 = max(bx0, 0), max(by0, 0), min(bx1, W - 1), min(by1, H - 1)
            if bx0 <= bx1 and by0 <= by1:
                scene[by0:by1 + 1, bx0:bx1 + 1] = col
    return scene


def render_device(frame, cmds: Sequence[tuple], stream=None):
    """The same raster on the MI355X (OMNI_OP_OVERLAY, csrc/overlay_png.hip): `frame` is a uint8 [H,W,3] device tensor, modified
    in place; the primitive table and the label masks (a few KB) are uploaded per call."""
    import torch
    from .. import _lib as L
    H, W = frame.shape[:2]
    P, M = raster_primitives(cmds)
    if P.shape[0] == 0:
        return frame
    assert frame.dtype == torch.uint8 and frame.is_contiguous() and frame.shape[2] == 3
    pd = torch.from_numpy(P).to(frame.device)
    md = torch.from_numpy(M).to(frame.device)
    L.launch(L.make_op(L.OP_OVERLAY, L.F32, p=[frame.data_ptr(), pd.data_ptr(), md.data_ptr()], i={0: H, 1: W, 2: P.shape[0]}), stream)
    return frame


class BoxAnnotator:
    """ref:util/box_annotator.py:10-44 constructor contract (colour palette fixed to the default one)."""

    def __init__(self, thickness: int = 3, text_scale: float = 0.5, text_thickness: int = 2, text_padding: int = 10,
                 avoid_overlap: bool = True):
        self.thickness, self.text_scale, self.text_thickness = thickness, text_scale, text_thickness
        self.text_padding, self.avoid_overlap = text_padding, avoid_overlap

    def plan(self, xyxy, labels, image_size):
        return plan_overlay(xyxy, labels, image_size, self.text_scale, self.text_padding, self.text_thickness, self.thickness,
                            self.avoid_overlap)

    def annotate(self, scene: np.ndarray, xyxy: np.ndarray, labels=None, image_size=None) -> np.ndarray:
        h, w = scene.shape[:2]
        return render(scene, self.plan(xyxy, labels, image_size or (w, h)))
