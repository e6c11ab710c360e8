"""Micro-benchmark of conv_igemm on the captioner's dominant GEMM shapes (HIP events, per variant).

VARIANTS="f32,split,dma,dma:256x128,dma:128x128" — kind[:tile] and optional +ENV=value pairs
applied for that variant only.  kind: f32 (exact f32 MFMA), split (register-staged split-f16), dma (pre-split LDS-DMA GEMM,
csrc/gemm_dma.hip: input converted to format B before the timed region, as its producer would have written it).  Accuracy is checked on 2048 rows sampled over the whole
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
    ("s0.qkv", 4718592, 384, 128, 0, False), ("s1.fc1", 1179648, 1024, 256, 2, False), ("s1.qkv", 1179648, 768, 256, 0, False),
]


def main():
    import torch
    from omniparser_amd import _lib as L
    from omniparser_amd.planner import PlanBuilder, View
    dtype = L.F32 if os.environ.get("OMNI_PRECISION", "f32") == "f32" else L.F16
    tdt = torch.float32 if dtype == L.F32 else torch.float16
    stream = torch.cuda.Stream()
    for variant in os.environ.get("VARIANTS", "split,dma,dma:256x128,dma:128x128").split(","):
        os.environ["OMNI_CONV_SPLIT"] = "0" if variant.startswith("f32") else "1"
        variant, *envs = variant.split("+")
        for k in ("OMNI_GEMM_TILE",):
            os.environ.pop(k, None)
        for kv in envs:
            k, v = kv.split("=")
            os.environ[k] = v
        parts = variant.split(":")
        dma = parts[0] == "dma"
        if len(parts) > 1 and dma:
            os.environ["OMNI_GEMM_TILE"] = parts[1]
        print(f"--- variant {variant} {' '.join(envs)}")
        only = [t for t in os.environ.get("SHAPES", "").split(",") if t]
        for name, M, N, K, act, res in SHAPES:
            if (only and name not in only) or (dma and act == 1):
                continue
            torch.cuda.synchronize()
            pb = PlanBuilder("cuda", dtype)
            x = View(torch.randn(1, M, 1, K, device="cuda").to(tdt), 0, K)
            xref = x.t.clone() if dma else x.t
            wcpu = torch.randn(N, K) * 0.05
            y = pb.alloc(1, M, 1, N)
            r = View(torch.randn(1, M, 1, N, device="cuda").to(tdt), 0, N) if res else None
            bias = torch.randn(N)
            torch.cuda.synchronize()
            if dma:
                pre = PlanBuilder("cuda", dtype)
                pre.split_convert(x)
                pre.build().run(stream); stream.synchronize()
                x.fmt = "split"
                w = pb.pack_weight_dma(wcpu)
                pb.conv(x, w, bias, y, 1, act=act, res=r, out_split=(act == 2))
            else:
                w = pb.pack_weight(wcpu[:, :, None, None])
                pb.conv(x, w, bias, y, 1, act=act, res=r)
            plan = pb.build()
            plan.run(stream); stream.synchronize()
            ms = plan.time(5, stream)
            # accuracy vs f64 on rows sampled over the whole M range (incl. the first and last tiles)
            rows = torch.cat([torch.arange(128), torch.arange(M - 128, M), torch.randint(0, M, (1792,))]).cuda()
            xs = xref[0, rows, 0, :].double().cpu()
            ref = xs @ wcpu.double().t()
            got = y.t[0, rows, 0, :].cpu()
            if dma and act == 2:            # format-B output: [C/16 groups][16 hi halves | 16 lo halves]
                hv = got.contiguous().view(torch.float16).view(got.shape[0], N // 16, 2, 16).double()
                got = (hv[:, :, 0] + hv[:, :, 1]).reshape(got.shape[0], N)
            got = got.double()
            ref = ref + bias.double()
            if act == 2:
                ref = torch.nn.functional.gelu(ref)
            elif act == 1:
                ref = torch.nn.functional.silu(ref)
            if res:
                ref = ref + r.t[0, rows, 0, :].double().cpu()
            err = ((got - ref).abs().max() / ref.abs().max()).item()
            print(f"{name:8s} M={M:8d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {2*M*N*K/ms/1e9:7.1f} TF/s  relerr {err:.2e}")
            del pb, plan, x, y, r
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
