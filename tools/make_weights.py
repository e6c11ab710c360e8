"""Create a stand-in `icon_detect_v3/model.pt` TorchScript blob with seeded random weights.

The reference downloads its detector blob from the Hugging Face hub (ref:util/yolov9.py:43-50); there
is no network and no checkpoint on this box, so benchmarks and tests feed the product a blob with the
same architecture (oracle/yolov9e_ref.py = public YOLOv9-E, 58.05 M params unfused) and the same
loading path (`torch.jit.load`).  This is input-data generation, not part of the product path.
"""
import argparse
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def make_blob(path, seed=0, nc=1, width=1.0):
    from oracle.yolov9e_ref import build_random_detector
    model = build_random_detector(seed=seed, nc=nc, width=width)
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with torch.no_grad():
        ts = torch.jit.trace(model, torch.rand(1, 3, 64, 64), check_trace=False)
    ts.save(str(path))
    return model


def default_path(seed=0, nc=1, width=1.0):
    tag = f"s{seed}_nc{nc}_w{width:g}"
    return ROOT / "weights" / f"icon_detect_v3_{tag}" / "icon_detect_v3" / "model.pt"


def ensure_blob(seed=0, nc=1, width=1.0):
    p = default_path(seed, nc, width)
    if not p.exists():
        make_blob(p, seed, nc, width)
    return p


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--nc", type=int, default=1)
    ap.add_argument("--width", type=float, default=1.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = a.out or default_path(a.seed, a.nc, a.width)
    make_blob(out, a.seed, a.nc, a.width)
    print(out)
