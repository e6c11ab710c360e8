"""How many of the synthetic frames does the MI355X reproduce ELEMENT FOR ELEMENT against the oracle's own list — measured, not
predicted from a-priori margins (tools/scan_parity_frames.py is the a-priori scan: 9 of 110 seeds have no NMS tie in the oracle)?

  python tools/scan_gpu_vs_oracle.py oracle 0:110    (any box; CPU only: the oracle's final boxes / scores per seed, full-width stand-in,
                                                      1920x1080 frames at 640x640 -> tools/_scan/oracle_finals.json; deterministic)
  python tools/scan_gpu_vs_oracle.py device          (GPU box: the detector's final boxes for the same seeds, then per frame: same count,
                                                      one-to-one pairing at IoU >= 0.999, same glue elements and crop rectangles)
The oracle is test infrastructure: this tool is a measurement script, nothing in the product imports it."""
import json
import sys
from pathlib import Path

import torch
from PIL import Image

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
CACHE = ROOT / "tools" / "_scan" / "oracle_finals.json"
IW, IH = 1920, 1080


def bits(t):
    return t.contiguous().view(torch.int32).flatten().tolist()


def unbits(v, cols=None):
    t = torch.tensor(v, dtype=torch.int32).view(torch.float32)
    return t.view(-1, cols) if cols else t


def run_oracle(lo, hi):
    from oracle import detector_ref as D
    from omniparser_amd.synth import synthetic_screenshot
    from tools.make_weights import ensure_blob
    m = torch.jit.load(str(ensure_blob(seed=0, nc=1, width=1.0)), map_location="cpu").eval()
    out = json.loads(CACHE.read_text()) if CACHE.exists() else {}
    for s in range(lo, hi):
        if str(s) in out:
            continue
        rb, rs, rc, dbg = D.predict(m, Image.fromarray(synthetic_screenshot(s, IW, IH)), conf=0.05, imgsz=640, iou=0.1, max_det=300, return_debug=True)
        out[str(s)] = {"boxes": bits(rb), "conf": bits(rs), "near_ties": int(dbg["near_ties"]), "score_ties": int(dbg["score_ties"]),
                       "candidates": int(dbg["valid"].sum())}
        print(s, len(rb), dbg["near_ties"], dbg["score_ties"], flush=True)
    CACHE.parent.mkdir(exist_ok=True)
    CACHE.write_text(json.dumps(out))


def provenance():
    """what the scan was measured WITH: bench.py quotes a scan only while these still describe the tree (the round-5 scan
    predated the conv tuning table, which changes the summation order of 84 conv shapes)."""
    import hashlib
    import subprocess
    def sha16(p):
        return hashlib.sha256(p.read_bytes()).hexdigest()[:16] if p.exists() else None
    try:
        rev = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
    except OSError:
        rev = None
    import os
    return {"git_rev": rev, "conv_tuning_sha16": sha16(ROOT / "omniparser_amd" / "conv_tuning_gfx950.json"),
            "conv_tuning_applied": os.environ.get("OMNI_CONV_TUNING", "1") != "0",
            "kernel_sources_sha16": hashlib.sha256(b"".join(p.read_bytes() for p in sorted((ROOT / "omniparser_amd" / "csrc").glob("*.h*")))).hexdigest()[:16]}


def run_device():
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob
    gold = json.loads(CACHE.read_text())
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=1.0), device="cuda", precision="f32")
    sp = ScreenParser(det, None, box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640)
    rows, n_exact, n_boxes_exact, tie_free, tie_free_exact, n_sets = [], 0, 0, 0, 0, 0
    for s in sorted(map(int, gold)):
        g = gold[str(s)]
        rb, rs = unbits(g["boxes"], 4), unbits(g["conf"])
        img = Image.fromarray(synthetic_screenshot(s, IW, IH))
        r = det.predict(img, conf=0.05, imgsz=640, iou=0.1)[0]
        gb, gs = r.boxes.xyxy.cpu(), r.boxes.conf.cpu()
        texts, obox = synthetic_ocr(s, IW, IH, 40)
        el_r, cr_r = sp.glue(rb, IW, IH, obox, texts)
        el_g, cr_g = sp.glue(gb, IW, IH, obox, texts)
        same_n = len(gb) == len(rb)
        boxes_ok = False
        if same_n and len(rb):
            # order-free one-to-one pairing at IoU >= 0.999 (two boxes whose scores agree to 1e-6 may exchange ranks)
            x1 = torch.maximum(rb[:, None, 0], gb[None, :, 0]); y1 = torch.maximum(rb[:, None, 1], gb[None, :, 1])
            x2 = torch.minimum(rb[:, None, 2], gb[None, :, 2]); y2 = torch.minimum(rb[:, None, 3], gb[None, :, 3])
            inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
            ar = (rb[:, 2] - rb[:, 0]) * (rb[:, 3] - rb[:, 1]); ag = (gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1])
            m = torch.nan_to_num(inter / (ar[:, None] + ag[None, :] - inter).clamp_min(1e-12), nan=1.0)
            best, arg = m.max(1)
            degenerate = (ar <= 0)                        # zero-area boxes have no IoU: matched by coordinates
            close = (rb[:, None, :] - gb[None, :, :]).abs().amax(2).min(1).values < 1e-2
            boxes_ok = bool(((best >= 0.999) | (degenerate & close)).all()) and len(set(arg[~degenerate].tolist())) == int((~degenerate).sum())
        elems_ok = boxes_ok and len(el_r) == len(el_g) and [list(c) for c in cr_r] == [list(c) for c in cr_g] and all(
            (a["type"], a["source"], a["interactivity"], a["content"]) == (b["type"], b["source"], b["interactivity"], b["content"]) and
            max(abs(x - y) for x, y in zip(a["bbox"], b["bbox"])) < 1e-5 for a, b in zip(el_g, el_r))
        tf = g["near_ties"] == 0 and g["score_ties"] == 0
        n_boxes_exact += boxes_ok; n_exact += elems_ok; tie_free += tf; tie_free_exact += (tf and elems_ok)
        row = {"seed": s, "boxes": [len(rb), len(gb)], "elements": [len(el_r), len(el_g)], "oracle_ties": [g["near_ties"], g["score_ties"]],
               "boxes_identical": bool(boxes_ok), "elements_identical": bool(elems_ok)}
        if boxes_ok and not elems_ok:
            # boxes pair one to one and the hand-off still differs (round 5: seed 63, unexplained): record WHICH decision — the first
            # element that differs, the largest box difference of the frame in pixels, and the hand-off's own margins for that element
            # (overlap-removal compares IoU / area ratios of ratio-coordinate boxes against fixed thresholds: a 1e-4 px difference can flip one)
            d = (rb[:, None, :] - gb[None, :, :]).abs().amax(2).min(1).values
            row["max_box_diff_px"] = float(d.max())
            for k, (a, b) in enumerate(zip(el_g, el_r)):
                if (a["type"], a["source"], a["content"]) != (b["type"], b["source"], b["content"]) or max(abs(x - y) for x, y in zip(a["bbox"], b["bbox"])) >= 1e-5:
                    row["first_difference"] = {"index": k, "device": {kk: a[kk] for kk in ("type", "source", "bbox")}, "oracle": {kk: b[kk] for kk in ("type", "source", "bbox")}}
                    break
            row["crop_rects_differ"] = [list(c) for c in cr_r] != [list(c) for c in cr_g]
            # round 6 (seeds 60 / 63 of the round-5 scan, "one hand-off decision differs"): both were RANK EXCHANGES — two non-overlapping
            # boxes whose oracle scores differ by 4e-7 / 6e-7 (an f32 ulp or two) leave the two NMS implementations in the other order, and the
            # element list follows the score order.  Same elements, same crop rectangles, as SETS; counted separately from real differences.
            left = list(el_g)
            same = len(el_g) == len(el_r)
            for b in el_r:                  # order-free pairing with the in-order comparison's own tolerance (1e-5 in ratio units)
                j = next((j for j, a in enumerate(left) if (a["type"], a["source"], a["interactivity"], a["content"]) ==
                          (b["type"], b["source"], b["interactivity"], b["content"]) and max(abs(x - y) for x, y in zip(a["bbox"], b["bbox"])) < 1e-5), None)
                if j is None:
                    same = False
                    break
                left.pop(j)
            row["identical_as_sets"] = bool(same and not left and sorted(map(tuple, cr_g)) == sorted(map(tuple, cr_r)))
            gaps = (rs[:-1] - rs[1:]).abs()
            row["oracle_adjacent_score_gaps_below_4e-6"] = int((gaps < 4e-6).sum())
            n_sets += bool(row["identical_as_sets"])
        rows.append(row)
    print(json.dumps({"provenance": provenance(), "frames": len(rows), "final_boxes_identical": n_boxes_exact, "elements_and_crops_identical": n_exact,
                      "identical_up_to_exchanges_of_equal_score_neighbours": n_exact + n_sets,
                      "oracle_tie_free_frames": tie_free, "tie_free_and_identical": tie_free_exact,
                      "definition": "identical = same count, one-to-one pairing at IoU >= 0.999 (zero-area boxes by coordinates), then the same element "
                                    "list (type / source / content / order, bbox within 1e-5 in ratio units) and the same integer crop rectangles",
                      "frames_not_identical": [r for r in rows if not r["elements_identical"]]}))


if __name__ == "__main__":
    if sys.argv[1] == "oracle":
        lo, hi = (int(v) for v in sys.argv[2].split(":"))
        run_oracle(lo, hi)
    else:
        run_device()
