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
    # NMS decisions of the per-tile passes and of the merge that sit on a tie, and the smallest margins (see detector_ref.nms / postprocess)
    stats = {"near_ties": 0, "score_ties": 0, "min_iou_margin": 1.0, "min_score_gap": 1.0, "thr_margin": 1.0}
    import math
    thr = math.log(conf / (1.0 - conf))
    for (x0, y0) in origins:
        b, s, c, dbg = D.predict(model, Image.fromarray(image_u8[y0:y0 + th, x0:x0 + tw]), conf=conf, imgsz=imgsz, iou=iou, max_det=max_det,
                                 return_debug=True)
        for k in ("near_ties", "score_ties"):
            stats[k] += int(dbg[k])
        for k in ("min_iou_margin", "min_score_gap"):
            stats[k] = min(stats[k], float(dbg[k]))
        if return_stats:
            with torch.inference_mode():
                out = model(dbg["input"])
            lg = torch.cat([out[i].flatten(2) for i in (0, 2, 4)], 2).max(1).values.flatten()
            stats["thr_margin"] = min(stats["thr_margin"], float((lg - thr).abs().min()))
        bs.append(b + torch.tensor([x0, y0, x0, y0], dtype=torch.float32)); ss.append(s); cs.append(c)
    b, s, c = torch.cat(bs), torch.cat(ss), torch.cat(cs)
    keep_all = D.batched_nms(b, s, c, iou)
    keep = keep_all[:max_det]
    merge = {"score_floor": float(s[keep[-1]]) - 1e-4 if len(keep_all) > max_det else -float("inf")}
    D.batched_nms(b, s, c, iou, merge, max_keep=max_det)          # tie statistics of the merge decisions that can reach the final list
    for k in ("near_ties", "score_ties"):
        stats[k] += int(merge.get(k, 0))
    for k in ("min_iou_margin", "min_score_gap"):
        stats[k] = min(stats[k], float(merge.get(k, 1.0)))
    b = b[keep].clone()
    b[:, [0, 2]] = b[:, [0, 2]].clamp(0, iw); b[:, [1, 3]] = b[:, [1, 3]].clamp(0, ih)
    if return_stats:
        return b, s[keep], c[keep], stats
    return b, s[keep], c[keep]
