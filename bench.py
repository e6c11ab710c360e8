#!/usr/bin/env python
"""bench.py — screenshots/sec of the MI355X screen-parsing hot path (BASELINE.json metric).

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one rank
per GPU).  A *step* is one synthetic 1920x1080 screenshot, already resident in HBM as RGB bytes,
through the hot path: Pillow-exact Lanczos letterbox -> YOLOv9-E -> DFL/decode/threshold -> NMS
(BASELINE config 2: detector only, batch 1; the captioner stage is reported once it exists).
W untimed warm-up steps, exactly K timed steps between barrier+synchronize pairs, MAX over ranks,
rank 0 prints ONE JSON line.  Screenshots shard across ranks with no data-path collective; the only
exchange is one all_gather of the packed element records at the end of the job (inside the timed
region).

Extra objects on the same line:
  roofline     — dominant kernel (conv_igemm_kernel family): algorithmic conv FLOPs per screenshot /
                 HIP-event time of the conv launches, vs the dense MFMA peak of the dtype.
  cpu_baseline — the reference-equivalent CPU path (oracle/detector_ref.py: torch.jit blob on host
                 cores + PIL letterbox + torch NMS) timed on rank 0 at N=1 on a bounded sample.
Weights are seeded-random (tools/make_weights.py); data is synthetic (omniparser_amd/synth.py).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default=os.environ.get("OMNI_PRECISION", "f32"), choices=["f32", "f16"])
    ap.add_argument("--imgsz", default="640", help="'640' (reference default) or 'native' (1088x1920 network input)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=3)
    ap.add_argument("--width", type=float, default=1.0, help="debug only: channel multiplier (1.0 = YOLOv9-E)")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from omniparser_amd import _lib as L
    from omniparser_amd import dist as OD
    from omniparser_amd.synth import synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob

    rank, world, local_rank = OD.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    IW, IH = 1920, 1080
    imgsz = 640 if args.imgsz == "640" else (IH, IW)
    conf, iou, max_det = 0.05, 0.1, 300     # ref:util/omniparser.py:30 / ref:util/utils.py:431

    if rank == 0:
        blob = ensure_blob(seed=0, nc=1, width=args.width)
    if world > 1:
        dist.barrier()
    blob = ensure_blob(seed=0, nc=1, width=args.width)
    det = YOLOv9Detector(model_path=blob, device=dev, precision=args.precision)
    dp = det.get_plan(IW, IH, imgsz, conf, iou, max_det, batch=1)

    # this rank's shard of the job's screenshots (8 distinct synthetic frames, cycled), resident in HBM
    n_total = (args.steps + args.warmup) * world
    frames = [torch.from_numpy(synthetic_screenshot(s, IW, IH)).to(dev) for s in range(8)]
    my_items = OD.shard_indices(args.steps * world, rank, world)
    assert len(my_items) == args.steps

    def step(item):
        with torch.cuda.stream(det.stream):
            dp.img[0].copy_(frames[item % 8], non_blocking=True)   # device->device, 6 MB
            dp.launch(det)

    def sync_all():
        det.stream.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for w in range(args.warmup):
        step(w)
    sync_all()

    recs = torch.zeros(len(my_items), OD.REC_W, dtype=torch.int32, device=dev)
    t0 = time.perf_counter()
    with torch.cuda.stream(det.stream):
        for li, item in enumerate(my_items):
            step(item)
            # results stay on device; pack without a host sync (count is clamped inside pack via K slot)
            recs[li, 0] = item
            recs[li, 1:2] = dp.out_count[0:1]
            o = 2
            recs[li, o:o + 4 * max_det] = dp.out_boxes[0].view(torch.int32).flatten()
            o += 4 * OD.MAX_DET
            recs[li, o:o + max_det] = dp.out_scores[0].view(torch.int32)
            o += OD.MAX_DET
            recs[li, o:o + max_det] = dp.out_cls[0]
        allr = OD.gather_records(recs, args.steps * world, rank, world)
    det.stream.synchronize()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_items = args.steps * world
    value = total_items / elapsed
    kept = allr[:, 1].float().mean().item()

    out = {
        "metric": "screenshots/sec end-to-end (detect+caption) @1920x1080",
        "value": round(value, 3),
        "unit": "screenshots/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic 1920x1080 GUI-like screenshots (8 seeds) + seeded random-weight YOLOv9-E blob",
        "config": {
            "workload": "BASELINE configs[1]: YOLOv9-E icon_detect only, batch=1, 1920x1080 "
                        "(letterbox+network+decode+NMS; captioner not yet in the timed path)",
            "network_input": "640x640" if args.imgsz == "640" else "1088x1920",
            "conf": conf, "iou": iou, "max_det": max_det,
            "parallelism": f"replicas x{world}, round-robin shards, 1 all_gather/job",
            "hipgraph": det.use_graph, "ops_per_screenshot": dp.n_ops,
            "mean_kept_boxes": round(kept, 2),
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel family, measured live with HIP events on det.stream
        conv_ops = [op for op in dp.plan.ops if op.kind == L.OP_CONV]
        conv_plan = L.Plan(conv_ops)
        conv_plan.run(det.stream); det.stream.synchronize()
        iters = 20
        conv_ms = conv_plan.time(iters, det.stream)
        full_ms = dp.plan.time(iters, det.stream)
        flops = dp.net_flops
        peak = 157.3 if args.precision == "f32" else 2500.0
        achieved = flops / (conv_ms * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "mfma", "kernel": "conv_igemm_kernel<T,BM,BN,ALIGNED> (all instantiations)",
            "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "traffic": None,
            "flops_per_screenshot": flops, "conv_launches": len(conv_ops),
            "conv_ms_per_screenshot": round(conv_ms, 4), "avg_launch_us": round(1000 * conv_ms / len(conv_ops), 3),
            "plan_ms_per_screenshot_hip_events": round(full_ms, 4),
            "algorithmic_bytes_per_screenshot": dp.net_bytes,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(blob, imgsz, conf, iou, max_det, args.cpu_samples)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(blob, imgsz, conf, iou, max_det, samples):
    """Reference-equivalent CPU path on the host cores (oracle restatement of ref:util/yolov9.py)."""
    import torch
    from PIL import Image
    from oracle import detector_ref as D
    from omniparser_amd.synth import synthetic_screenshot
    model = torch.jit.load(str(blob), map_location="cpu").eval()
    imgs = [Image.fromarray(synthetic_screenshot(s)) for s in range(samples)]
    D.predict(model, imgs[0], conf=conf, imgsz=imgsz, iou=iou, max_det=max_det)   # warm-up
    t0 = time.perf_counter()
    for im in imgs:
        D.predict(model, im, conf=conf, imgsz=imgsz, iou=iou, max_det=max_det)
    dt = time.perf_counter() - t0
    return {"value": round(samples / dt, 4), "unit": "screenshots/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{samples} synthetic 1920x1080 screenshots, detector stage "
            f"(PIL Lanczos letterbox + TorchScript YOLOv9-E fp32 on CPU + decode + batched_nms), 1 warm-up"}


if __name__ == "__main__":
    main()
