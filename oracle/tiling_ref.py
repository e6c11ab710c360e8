"""ORACLE (test infrastructure): CPU statement of the tiled-detection policy used for >1080p frames
(BASELINE configs[4]).  The policy does not exist in the reference (it letterboxes the whole image —
SURVEY 0.7); this file only states OUR merge policy for given tiles: reference detector per tile
(oracle.detector_ref.predict) -> shift -> global batched_nms(iou)[:max_det] -> clamp."""
import numpy as np
import torch
from PIL import Image

from . import detector_ref as D


def predict_tiled(model, image_u8: np.ndarray, origins, tw, th, conf=0.05, imgsz=640, iou=0.1, max_det=300, return_stats=False):
    """origins: [(x0, y0)] of the tw x th tiles — the tiling geometry is an input here (tests pin the product's
    geometry with literal expected values instead of comparing two copies of one function)."""
    ih, iw = image_u8.shape[:2]
    bs, ss, cs = [], [], []
    stats = {}                        # NMS decisions of the per-tile passes and of the merge that sit on a tie (see detector_ref.nms)
    for (x0, y0) in origins:
        b, s, c, dbg = D.predict(model, Image.fromarray(image_u8[y0:y0 + th, x0:x0 + tw]), conf=conf, imgsz=imgsz, iou=iou, max_det=max_det,
                                 return_debug=True)
        for k in ("near_ties", "score_ties"):
            stats[k] = stats.get(k, 0) + int(dbg[k])
        bs.append(b + torch.tensor([x0, y0, x0, y0], dtype=torch.float32)); ss.append(s); cs.append(c)
    b, s, c = torch.cat(bs), torch.cat(ss), torch.cat(cs)
    keep = D.batched_nms(b, s, c, iou, stats)[:max_det]
    b = b[keep].clone()
    b[:, [0, 2]] = b[:, [0, 2]].clamp(0, iw); b[:, [1, 3]] = b[:, [1, 3]].clamp(0, ih)
    if return_stats:
        return b, s[keep], c[keep], stats
    return b, s[keep], c[keep]
