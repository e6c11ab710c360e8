"""BASELINE configs[3] stand-in: a ScreenSpot-Pro-sized eval set of mixed-resolution screenshots streamed through
`omniparser_amd.stream.run_stream` on the GPUs of one node (one process per GPU; run under torchrun for N > 1):

    python tools/stream_bench.py --items 64 --caption-res 64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/stream_bench.py --items 1581
    python tools/stream_bench.py --gpus 8 --items 1581          # the same job, self-launched (omniparser_amd.dist.self_launch)

The dataset is absent (no network): frames are synthetic screenshots at the resolution mix of `stream.RESOLUTION_MIX`
(documented as synthetic), a small pool per resolution kept resident in HBM, OCR boxes synthetic.  Prints one JSON
line on rank 0: screenshots/s over the whole job (max over ranks), batches, per-resolution counts, gathered records
checked for completeness.  Not the headline bench (bench.py is); never run on a GPU in round 1."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--caption-res", type=int, default=768, choices=[64, 768])
    ap.add_argument("--pool", type=int, default=2, help="distinct synthetic frames kept per resolution")
    ap.add_argument("--ocr", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=None, help="ranks; with no launcher environment (WORLD_SIZE unset) and N > 1 the script spawns them itself")
    a = ap.parse_args()
    if a.gpus and a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from omniparser_amd.dist import self_launch
        sys.exit(self_launch(a.gpus))
    import torch
    import torch.distributed as dist
    from omniparser_amd import dist as OD
    from omniparser_amd import stream as ST
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import caption_dir, default_path, ensure_via_subprocess

    rank, world, local_rank = OD.init_from_env()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if rank == 0:
        ensure_via_subprocess("detector", seed=0, nc=1, width=1.0)
        ensure_via_subprocess("caption", seed=0)
    if world > 1:
        dist.barrier()
    det = YOLOv9Detector(model_path=default_path(0, 1, 1.0), device=dev)
    cap = Florence2Captioner(caption_dir(0), dev, resolution=a.caption_res)
    parser = ScreenParser(det, cap, box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640)
    sizes = ST.synthetic_sizes(a.items, seed=a.seed)
    pool = {}
    for (w, h) in sorted(set(sizes)):
        pool[(w, h)] = [(torch.from_numpy(synthetic_screenshot(1000 * k + w % 997, w, h)).to(dev), synthetic_ocr(k, w, h, a.ocr))
                        for k in range(a.pool)]

    def load(i):
        frames = pool[tuple(sizes[i])]
        return frames[i % len(frames)]

    os.environ.setdefault("OMNI_MAX_DETECT_PLANS", "16")        # one plan per resolution stays resident

    def parse(frames, ocr):
        return parser.parse_batch(frames, ocr, return_ids=True, pad_to=a.batch)   # ragged batches reuse the full-batch plan

    # warm-up: one batch per resolution builds (and captures) its detector plan outside the timed region
    for (w, h) in pool:
        parse([pool[(w, h)][0][0]], [pool[(w, h)][0][1]])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    out = ST.run_stream(sizes, load, parse, rank=rank, world=world, batch=a.batch, chunk=a.chunk)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rec = out["records"]
    assert rec.shape[0] == a.items and bool((rec[:, 0] == torch.arange(a.items, dtype=torch.int32)).all()), "gathered records incomplete"
    if rank == 0:
        counts = {}
        for s in sizes:
            counts[f"{s[0]}x{s[1]}"] = counts.get(f"{s[0]}x{s[1]}", 0) + 1
        print(json.dumps({
            "metric": "screenshots/sec, mixed-resolution eval stream (BASELINE configs[3] stand-in)", "value": round(a.items / elapsed, 4),
            "unit": "screenshots/s", "n_gpus": world, "items": a.items, "seconds": round(elapsed, 3), "caption_res": a.caption_res,
            "device_batch": a.batch, "gather_chunk": a.chunk, "rank0_batches": out["batches"], "rank0_items": out["items"],
            "mean_elements": round(float(rec[:, 1].float().mean()), 2), "resolution_counts": counts,
            "data": "synthetic screenshots at stream.RESOLUTION_MIX (dataset absent), seeded random-weight models"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
