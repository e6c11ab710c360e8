"""Micro-benchmark of conv_igemm on the captioner's dominant GEMM shapes (HIP events, per variant).

VARIANTS="f32,split:128x128:2,split:128x128:2+OMNI_XCD_NSPLIT=0,split:128x128:4" — kind[:tile[:split variant]] and
optional +ENV=value pairs applied for that variant only.  Accuracy is checked on 2048 rows sampled over the whole
M range (a wrong block -> tile permutation would leave rows unwritten or doubly written)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

SHAPES = [  # name, M, N, K, act, res
    ("s2.fc1", 294912, 2048, 512, 2, False), ("s2.fc2", 294912, 512, 2048, 0, True), ("s2.qkv", 294912, 1536, 512, 0, False),
    ("s2.proj", 294912, 512, 512, 0, True), ("s0.fc1", 4718592, 512, 128, 2, False), ("s0.fc2", 4718592, 128, 512, 0, True),
    ("enc.fc1", 74880, 3072, 768, 2, False), ("s3.fc1", 73728, 4096, 1024, 2, False), ("det.p3", 51200, 128, 1152, 1, False),
]


def main():
    import torch
    from omniparser_amd import _lib as L
    from omniparser_amd.planner import PlanBuilder, View
    dtype = L.F32 if os.environ.get("OMNI_PRECISION", "f32") == "f32" else L.F16
    tdt = torch.float32 if dtype == L.F32 else torch.float16
    stream = torch.cuda.Stream()
    for variant in os.environ.get("VARIANTS", "f32,split:128x64,split:128x128").split(","):
        os.environ["OMNI_CONV_SPLIT"] = "1" if variant.startswith("split") else "0"
        variant, *envs = variant.split("+")
        for k in ("OMNI_XCD_NSPLIT", "OMNI_XCD_L2_BUDGET_KB"):
            os.environ.pop(k, None)
        for kv in envs:
            k, v = kv.split("=")
            os.environ[k] = v
        parts = variant.split(":")
        if len(parts) > 1:
            os.environ["OMNI_SPLIT_TILE"] = parts[1]
        os.environ["OMNI_SPLIT_VARIANT"] = parts[2] if len(parts) > 2 else "0"
        print(f"--- variant {variant} {' '.join(envs)}")
        for name, M, N, K, act, res in SHAPES:
            pb = PlanBuilder("cuda", dtype)
            x = View(torch.randn(1, M, 1, K, device="cuda").to(tdt), 0, K)
            wcpu = torch.randn(N, K) * 0.05
            w = pb.pack_weight(wcpu[:, :, None, None])
            y = pb.alloc(1, M, 1, N)
            r = View(torch.randn(1, M, 1, N, device="cuda").to(tdt), 0, N) if res else None
            bias = torch.randn(N)
            pb.conv(x, w, bias, y, 1, act=act, res=r)
            plan = pb.build()
            plan.run(stream); stream.synchronize()
            ms = plan.time(5, stream)
            # accuracy vs f64 on rows sampled over the whole M range (incl. the first and last tiles)
            rows = torch.cat([torch.arange(128), torch.arange(M - 128, M), torch.randint(0, M, (1792,))]).cuda()
            xs = x.t[0, rows, 0, :].double().cpu()
            ref = xs @ wcpu.double().t()
            got = y.t[0, rows, 0, :].double().cpu()
            if act == 0 and not res:
                err = ((got - (ref + bias.double())).abs().max() / ref.abs().max()).item()
            else:
                err = float("nan")
            print(f"{name:8s} M={M:8d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {2*M*N*K/ms/1e9:7.1f} TF/s  relerr {err:.2e}")
            del pb, plan, x, y, r
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
