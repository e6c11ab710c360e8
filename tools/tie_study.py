"""How much does the CPU ORACLE's own result move when its candidates move by what a second f32 implementation differs by?
Perturbs the oracle's candidate scores (+-1e-6) and boxes (+-2e-4 px) — the measured MI355X-vs-oracle differences — and re-runs the
restated batched_nms[:300] + clamp + the detect->caption hand-off; prints, per frame, the distribution of (boxes without a twin,
box-count change, elements without a twin, element-count change) over the trials.  Source of the last section of
profiles/r2_parity_frame_scan.md and of the tie rules of tests/gpu_checks.py (assert_detector_frame, check_bench_path).  CPU only:
    python tools/tie_study.py --width 1.0 --frames 2 4 6 --trials 40"""
import argparse
import sys
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=float, default=1.0)
    ap.add_argument("--frames", type=int, nargs="+", default=[2, 4, 6])
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--score-noise", type=float, default=1e-6)
    ap.add_argument("--box-noise", type=float, default=2e-4)
    a = ap.parse_args()
    import torch
    from PIL import Image
    import gpu_checks as G
    from oracle import detector_ref as D
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from tools.make_weights import ensure_blob
    model = torch.jit.load(str(ensure_blob(0, 1, a.width)), map_location="cpu").eval()
    sp = ScreenParser(None, None, processor=object(), box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640)
    IW, IH = 1920, 1080
    for f in a.frames:
        img = synthetic_screenshot(f, IW, IH)
        texts, obox = synthetic_ocr(f, IW, IH, 40)
        rb, rs, rc, dbg = D.predict(model, Image.fromarray(img), conf=0.05, imgsz=640, iou=0.1, max_det=300, return_debug=True)
        el_r, _ = sp.glue(rb, IW, IH, obox, texts)
        cb, cs, cc = dbg["cand"]
        rbx = torch.tensor([e["bbox"] for e in el_r]).reshape(-1, 4)
        g = torch.Generator().manual_seed(0)
        res = []
        for _ in range(a.trials):
            ps = cs + (torch.rand(cs.shape, generator=g) - 0.5) * 2 * a.score_noise
            pb = cb + (torch.rand(cb.shape, generator=g) - 0.5) * 2 * a.box_noise
            xb, xs, xc = G._oracle_nms_clamp(pb, ps, cc, 0.1, 300, IW, IH)
            el_x, _ = sp.glue(xb, IW, IH, obox, texts)
            gbx = torch.tensor([e["bbox"] for e in el_x]).reshape(-1, 4)
            res.append((int((G.box_similarity(rb, xb).max(1).values < 0.999).sum()), len(xb) - len(rb),
                        int((G.ratio_box_iou(rbx, gbx).max(1).values < 0.999).sum()), len(el_x) - len(el_r)))
        print(f"frame {f}: near ties {int(dbg['near_ties'])}, score ties {int(dbg['score_ties'])}, {len(rb)} boxes, {len(el_r)} elements: "
              f"{Counter(res).most_common(6)}")


if __name__ == "__main__":
    main()
