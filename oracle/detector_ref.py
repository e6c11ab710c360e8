"""ORACLE (test infrastructure; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this).

CPU fp32 restatement of the reference detector adapter, ref:util/yolov9.py:52-136, plus a pure-torch
restatement of torchvision.ops.batched_nms / nms (torchvision is absent here and is unpinned in
ref:requirements.txt:3 — algorithm restated from its published CPU source).

The batched_nms / nms restatement itself is PARITY UNPINNED against torchvision (absent here); tests/test_third_party_pins_cpu.py holds the
pin that activates by itself where torchvision is importable (clouds below and above the 4000-numel switch, exact score ties).

Pinning: the reference has no tests for this path ("parity unpinned" by its own fixtures).  This
restatement is pinned instead against the reference's literal YOLOv9Detector class executed from
/root/reference under dependency shims (tests/golden/gen_golden.py, fixtures in tests/golden/).
"""
from contextlib import nullcontext

import numpy as np
import torch
from PIL import Image

STRIDES = (8, 16, 32)


# ------------------------------------------------------------------ torchvision.ops restatement
def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float, stats: dict = None, max_keep: int = None) -> torch.Tensor:
    """torchvision nms_kernel_impl (CPU): stable descending sort, greedy, strict `>`.
    max_keep (diagnostic passes only): stop after that many keeps — later iterations cannot change earlier keeps.
    stats["score_floor"] (optional, set by postprocess): only decisions about victims whose score reaches the floor are counted — a box
    scoring below the max_det-th keep can never enter the final list, whichever way a tie about it falls.
    stats (test diagnostics): counts the suppression decisions whose IoU lies within 1e-5 of the threshold ("near_ties") — those
    are the decisions a 1e-6 perturbation of the boxes can flip, in ANY implementation (the oracle's own f64 run included) — and the
    suppressions of a live box by a kept box whose score is no more than 4e-6 higher ("score_ties": with the two scores exchanged the
    other box of the pair survives; flat GUI regions give many anchors identical logits, so these are common on synthetic frames)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().cpu().numpy().astype(np.float32)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    order = torch.sort(scores.detach().cpu(), stable=True, descending=True).indices.numpy()
    sc = scores.detach().cpu().numpy().astype(np.float32)
    n = len(order)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    zero = np.float32(0)
    floor = np.float32(stats.get("score_floor", -np.inf)) if stats is not None else None
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(zero, xx2 - xx1)
        h = np.maximum(zero, yy2 - yy1)
        inter = (w * h).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / ((areas[i] + areas[rest]).astype(np.float32) - inter)
        if stats is not None:
            live = ~suppressed[rest] & (sc[rest] >= floor)
            stats["near_ties"] = stats.get("near_ties", 0) + int((np.abs(ovr[live] - thr) <= 1e-5).sum())
            hit = live & (ovr > thr)
            stats["score_ties"] = stats.get("score_ties", 0) + int((hit & (sc[i] - sc[rest] <= 4e-6)).sum())
            if live.any():
                stats["min_iou_margin"] = min(stats.get("min_iou_margin", 1.0), float(np.abs(ovr[live] - thr).min()))
            if hit.any():
                stats["min_score_gap"] = min(stats.get("min_score_gap", 1.0), float((sc[i] - sc[rest][hit]).min()))
        suppressed[rest[ovr > thr]] = True
        if max_keep is not None and len(keep) >= max_keep:
            break
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def batched_nms(boxes, scores, idxs, iou_threshold, stats: dict = None, max_keep: int = None):
    """torchvision.ops.batched_nms with the CPU dispatch threshold (numel > 4000 -> per-class loop)."""
    if boxes.numel() > 4000:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for class_id in torch.unique(idxs):
            curr = torch.where(idxs == class_id)[0]
            k = nms(boxes[curr], scores[curr], iou_threshold, stats, max_keep)
            keep_mask[curr[k]] = True
        keep_indices = torch.where(keep_mask)[0]
        return keep_indices[scores[keep_indices].sort(descending=True, stable=True)[1]]
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold, stats, max_keep)


# ------------------------------------------------------------------ ref:util/yolov9.py restated
def normalize_image_size(image_size):
    """ref:util/yolov9.py:52-62."""
    if isinstance(image_size, int):
        width = height = image_size
    elif len(image_size) == 2:
        height, width = image_size
    else:
        raise ValueError(f"Expected one or two image dimensions, got {image_size}")
    return ((int(width) + 31) // 32) * 32, ((int(height) + 31) // 32) * 32


def load_image(source):
    """ref:util/yolov9.py:64-71."""
    if isinstance(source, Image.Image):
        return source.convert("RGB")
    if isinstance(source, np.ndarray):
        return Image.fromarray(source).convert("RGB")
    with Image.open(source) as image:
        return image.convert("RGB")


def letterbox_geometry(image_width, image_height, image_size):
    """scale / resized dims / pads exactly as ref:util/yolov9.py:74-80."""
    tw, th = normalize_image_size(image_size)
    scale = min(tw / image_width, th / image_height)
    rw, rh = int(image_width * scale), int(image_height * scale)
    return tw, th, scale, rw, rh, (tw - rw) // 2, (th - rh) // 2


def preprocess(image: Image.Image, image_size):
    """ref:util/yolov9.py:73-87 (PIL LANCZOS letterbox, /255)."""
    tw, th, scale, rw, rh, pad_left, pad_top = letterbox_geometry(image.width, image.height, image_size)
    resized = image.resize((rw, rh), Image.Resampling.LANCZOS)
    padded = Image.new("RGB", (tw, th), (114, 114, 114))
    padded.paste(resized, (pad_left, pad_top))
    arr = np.asarray(padded, dtype=np.float32).transpose(2, 0, 1) / 255.0
    return torch.from_numpy(arr).unsqueeze(0), scale, pad_left, pad_top


def decode(outputs):
    """ref:util/yolov9.py:89-108."""
    class_logits, decoded = [], []
    for oi, stride in zip(range(0, len(outputs), 2), STRIDES):
        logits, dist = outputs[oi:oi + 2]
        bsz, _, h, w = logits.shape
        logits = logits.permute(0, 2, 3, 1).reshape(bsz, -1, logits.shape[1])
        dist = dist.permute(0, 2, 3, 1).reshape(bsz, -1, 4) * stride
        gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        anchors = (torch.stack((gx, gy), dim=-1).reshape(-1, 2) + 0.5) * stride
        lt, rb = dist.chunk(2, dim=-1)
        decoded.append(torch.cat((anchors - lt, anchors + rb), dim=-1))
        class_logits.append(logits)
    return torch.cat(class_logits, dim=1).sigmoid(), torch.cat(decoded, dim=1)


def postprocess(outputs, image_width, image_height, scale, pad_left, pad_top, conf, iou, max_det):
    """ref:util/yolov9.py:121-136 after the network call: decode, max-class, threshold, un-letterbox,
    batched_nms[:max_det], clamp.  Returns (boxes, scores, class_ids, debug)."""
    class_scores, boxes = decode(outputs)
    scores, class_ids = class_scores[0].max(dim=-1)
    valid = scores > conf
    scores, class_ids, boxes = scores[valid], class_ids[valid], boxes[0][valid]
    boxes[:, [0, 2]] = (boxes[:, [0, 2]] - pad_left) / scale
    boxes[:, [1, 3]] = (boxes[:, [1, 3]] - pad_top) / scale
    cand = (boxes.clone(), scores.clone(), class_ids.clone())
    keep_all = batched_nms(boxes, scores, class_ids, iou)                 # the reference's call, unabridged
    keep = keep_all[:max_det]
    # tie statistics of the decisions that can reach the FINAL list (test diagnostics; a second, abridged pass): the final list is the
    # first max_det keeps in score order, so (a) nothing decided after the max_det-th keep and (b) nothing about a victim scoring below
    # that keep (minus a 1e-4 guard band) can change it.  With fewer than max_det keeps (every 640x640 frame) this counts every
    # decision, as rounds 2-4 did; at 1088x1920 (~9 000 candidates, ~1 500 keeps) it drops the ~97 % of decisions that cannot matter.
    nms_stats = {"score_floor": float(scores[keep[-1]]) - 1e-4 if len(keep_all) > max_det else -float("inf")}
    batched_nms(boxes, scores, class_ids, iou, nms_stats, max_keep=max_det)
    boxes, scores, class_ids = boxes[keep], scores[keep], class_ids[keep]
    boxes[:, [0, 2]] = boxes[:, [0, 2]].clamp(0, image_width)
    boxes[:, [1, 3]] = boxes[:, [1, 3]].clamp(0, image_height)
    return boxes, scores, class_ids, {"cand": cand, "valid": valid, "keep": keep, "near_ties": nms_stats.get("near_ties", 0),
                                     "score_ties": nms_stats.get("score_ties", 0), "min_iou_margin": nms_stats.get("min_iou_margin", 1.0),
                                     "min_score_gap": nms_stats.get("min_score_gap", 1.0)}


@torch.inference_mode()
def predict(model, source, conf=0.25, imgsz=640, iou=0.7, max_det=300, return_debug=False):
    """ref:util/yolov9.py:115-136.  Returns (boxes[K,4], scores[K], class_ids[K])."""
    image = load_image(source)
    x, scale, pad_left, pad_top = preprocess(image, imgsz)
    boxes, scores, class_ids, dbg = postprocess(model(x), image.width, image.height, scale, pad_left, pad_top, conf, iou, max_det)
    if return_debug:
        dbg.update(input=x, scale=scale, pad_left=pad_left, pad_top=pad_top)
        return boxes, scores, class_ids, dbg
    return boxes, scores, class_ids
