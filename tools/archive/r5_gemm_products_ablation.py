"""Products-per-MAC ablation of the LDS-DMA GEMM (round 5): what does the third f16 product of every MAC buy, and
what would dropping it cost?  For OMNI_GEMM_PRODUCTS in {7 (shipped: hi*hi + w_lo*x_hi + w_hi*x_lo), 5 (no w_lo*x_hi: weights at f16
precision), 3 (no w_hi*x_lo: activations at f16 precision)} x OMNI_GEMM_PRODUCTS_MINK in {0 (every 256x256-tile layer), 2048 (the fc2
layers of DaViT stage 2 / the encoder only)}: the caption ids of every crop of 16 screenshots (the 8 benched frames + seeds 0..7,
~700 crops at 768x768) against the shipped composition's ids (which the GPU suite holds token-exact against transformers), and the
GEMM-family time of the bench step.  The knob exists in the library for this tool only; the product never sets it.

  python tools/gemm_products_ablation.py            -> one JSON line per setting + a summary line (gpurun_out/r5_products/*.json)
  python tools/gemm_products_ablation.py --inner    (one setting, environment already set: ids + timing)"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
SETTINGS = [(7, 0), (5, 0), (3, 0), (5, 2048), (3, 2048)]


def inner():
    import torch
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import BENCH_SEEDS, synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import caption_dir, default_path, ensure_via_subprocess
    ensure_via_subprocess("detector", seed=0, nc=1, width=1.0)
    ensure_via_subprocess("caption", seed=0)
    dev = torch.device("cuda", 0)
    det = YOLOv9Detector(model_path=default_path(0, 1, 1.0), device=dev, precision="f32")
    cap = Florence2Captioner(caption_dir(0), dev, precision="f32", resolution=768)
    sp = ScreenParser(det, cap, box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640)
    out = {"products": int(os.environ.get("OMNI_GEMM_PRODUCTS", "7")), "min_k": int(os.environ.get("OMNI_GEMM_PRODUCTS_MINK", "0")), "ids": [],
           "batch_ms": []}
    with torch.inference_mode():
        for seeds in (BENCH_SEEDS, tuple(range(8))):
            frames = [torch.from_numpy(synthetic_screenshot(s, 1920, 1080)).to(dev) for s in seeds]
            ocr = [synthetic_ocr(s, 1920, 1080, 40) for s in seeds]
            sp.parse_batch(frames, ocr, return_ids=True)                    # warm-up: plans built and captured
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            elems, ids = sp.parse_batch(frames, ocr, return_ids=True)
            torch.cuda.synchronize(dev)
            out["batch_ms"].append(round(1000 * (time.perf_counter() - t0), 2))
            out["ids"] += [[r.tolist() for r in f] for f in ids]
        # GEMM-family device time of one 128-row encode plan (HIP events around every op of an eager replay)
        cp = cap.plans(128, 768, 20)
        ms = cp.encode_plan.profile(cap.stream)
        out["encode128_gemm_ms"] = round(sum(t for op, t in zip(cp.encode_plan.ops, ms) if op.kind in (1, 24)), 3)
        out["encode128_all_ms"] = round(sum(ms), 3)
    print(json.dumps(out), flush=True)


def main():
    if "--inner" in sys.argv:
        return inner()
    od = ROOT / "gpurun_out" / "r5_products"
    od.mkdir(parents=True, exist_ok=True)
    res = {}
    for prod, mink in SETTINGS:
        env = dict(os.environ, OMNI_GEMM_PRODUCTS=str(prod), OMNI_GEMM_PRODUCTS_MINK=str(mink))
        try:
            r = subprocess.run([sys.executable, __file__, "--inner"], env=env, capture_output=True, text=True, timeout=420)
        except subprocess.TimeoutExpired:
            print(json.dumps({"products": prod, "min_k": mink, "error": "timeout"})); continue
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode or not lines:
            print(json.dumps({"products": prod, "min_k": mink, "error": r.stderr[-400:]})); continue
        d = json.loads(lines[-1])
        (od / f"prod{prod}_mink{mink}.json").write_text(json.dumps(d))
        res[(prod, mink)] = d
    base = res.get((7, 0))
    summary = []
    for (prod, mink), d in res.items():
        row = {"products": prod, "min_k": mink, "batch_ms": d["batch_ms"], "encode128_gemm_ms": d["encode128_gemm_ms"], "encode128_all_ms": d["encode128_all_ms"]}
        if base is not None:
            flat_b = [r for f in base["ids"] for r in f]
            flat_d = [r for f in d["ids"] for r in f]
            row["crops"] = len(flat_b)
            row["crops_with_different_ids"] = sum(a != b for a, b in zip(flat_b, flat_d)) + abs(len(flat_b) - len(flat_d))
        summary.append(row)
        print(json.dumps(row), flush=True)
    (od / "summary.json").write_text(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
