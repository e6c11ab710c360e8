"""ORACLE (test infrastructure): CPU statement of the tiled-detection policy used for >1080p frames
(BASELINE configs[4]).  The policy does not exist in the reference (it letterboxes the whole image —
SURVEY 0.7); this file only pins OUR policy: overlapping tiles -> reference detector per tile
(oracle.detector_ref.predict) -> shift -> global batched_nms(iou)[:max_det] -> clamp."""
import numpy as np
import torch
from PIL import Image

from . import detector_ref as D


def tile_origins(iw, ih, tile_w=1952, tile_h=1112, overlap=64):
    def axis(n, t):
        if n <= t:
            return [0]
        k = -(-(n - overlap) // (t - overlap))
        step = (n - t) / (k - 1)
        return [int(round(i * step)) for i in range(k)]
    return [(x, y) for y in axis(ih, tile_h) for x in axis(iw, tile_w)], min(tile_w, iw), min(tile_h, ih)


def predict_tiled(model, image_u8: np.ndarray, conf=0.05, imgsz=640, iou=0.1, max_det=300):
    ih, iw = image_u8.shape[:2]
    origins, tw, th = tile_origins(iw, ih)
    bs, ss, cs = [], [], []
    for (x0, y0) in origins:
        b, s, c = D.predict(model, Image.fromarray(image_u8[y0:y0 + th, x0:x0 + tw]), conf=conf, imgsz=imgsz, iou=iou, max_det=max_det)
        bs.append(b + torch.tensor([x0, y0, x0, y0], dtype=torch.float32)); ss.append(s); cs.append(c)
    b, s, c = torch.cat(bs), torch.cat(ss), torch.cat(cs)
    keep = D.batched_nms(b, s, c, iou)[:max_det]
    b = b[keep].clone()
    b[:, [0, 2]] = b[:, [0, 2]].clamp(0, iw); b[:, [1, 3]] = b[:, [1, 3]].clamp(0, ih)
    return b, s[keep], c[keep]
