"""Scan synthetic frames for the properties the box-for-box parity tests need (CPU only, oracle only):
  noise      max |f32 - f64| over the stand-in's head tensors (conditioning of the oracle on that frame)
  thr_margin smallest distance of an anchor's best class logit from logit(conf) (a candidate cannot flip below GPU rounding noise)
  near_ties / score_ties / iou_margin / score_gap   NMS decisions of the oracle that sit on a tie (oracle/detector_ref.py::nms)
Frames with noise <= 3.3e-5, no ties and comfortable margins go into tools/make_weights.py::EXACT_FRAMES.
usage: python tools/scan_parity_frames.py [width ...]   -> markdown table on stdout (committed as profiles/r2_parity_frame_scan.md)"""
import math
import os
import sys

import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import detector_ref as D                      # noqa: E402
from oracle.yolov9e_ref import YOLOv9E                    # noqa: E402
from omniparser_amd.synth import synthetic_screenshot     # noqa: E402
from tools.make_weights import ensure_blob                # noqa: E402

CASES = {1.0: [(s, 640, 1920, 1080) for s in range(8)],
         0.5: [(s, 640, 1920, 1080) for s in range(8)],
         0.25: [(s, 640, 1920, 1080) for s in range(4)] + [(s, 320, 640, 480) for s in range(4)] + [(0, (1080, 1920), 1920, 1080)]}


def main():
    # `--seeds A:B` scans full-HD frames A..B-1 at 640x640 for the widths given (choosing the bench's tie-free frames)
    args = sys.argv[1:]
    if "--seeds" in args:
        k = args.index("--seeds")
        lo, hi = (int(v) for v in args[k + 1].split(":"))
        del args[k:k + 2]
        for w in (float(a) for a in args) if args else (1.0,):
            CASES[w] = [(s, 640, 1920, 1080) for s in range(lo, hi)]
    widths = [float(a) for a in args] or [0.25, 0.5, 1.0]
    thr = math.log(0.05 / 0.95)
    print("| width | seed | frame | net input | boxes | candidates | noise f32-f64 | thr margin | near ties | score ties | IoU margin | score gap |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for width in widths:
        m = torch.jit.load(str(ensure_blob(seed=0, nc=1, width=width)), map_location="cpu").eval()
        m64 = YOLOv9E(nc=1, width=width).double()
        m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()}, strict=False)
        m64.eval()
        for seed, imgsz, iw, ih in CASES[width]:
            img = Image.fromarray(synthetic_screenshot(seed, iw, ih))
            rb, rs, rc, dbg = D.predict(m, img, conf=0.05, imgsz=imgsz, iou=0.1, return_debug=True)
            x = dbg["input"]
            with torch.inference_mode():
                o32 = m(x)
                o64 = m64(x.double())
            noise = max((a.double() - b).abs().max().item() for a, b in zip(o32, o64))
            lg = torch.cat([o32[i].flatten(2) for i in (0, 2, 4)], 2).max(1).values.flatten()
            print(f"| {width} | {seed} | {iw}x{ih} | {tuple(x.shape[2:])} | {len(rb)} | {int(dbg['valid'].sum())} | {noise:.2e} | "
                  f"{(lg - thr).abs().min().item():.2e} | {dbg['near_ties']} | {dbg['score_ties']} | {dbg['min_iou_margin']:.2e} | "
                  f"{dbg['min_score_gap']:.2e} |", flush=True)


if __name__ == "__main__":
    main()
