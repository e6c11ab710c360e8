"""Power / clock samples of the GPU while one plan replays in a loop: `rocm-smi` polled from a thread every ~0.25 s while (a) the
long-K LDS-DMA GEMM (DaViT stage-2 fc2: M = 294912, N = 512, K = 2048), (b) the GELU GEMM (fc1: N = 2048, K = 512) and (c) an HBM-bound
kernel (LayerNorm over 4.7 M x 128) each run back to back for a few seconds — evidence for "the GEMM family runs at the board's power
limit with the clock pulled down, the HBM-bound kernels do not" (DESIGN 5).
usage (GPU box): python tools/power_trace.py > gpurun_out/.../power_trace.json"""
import json
import re
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def smi_sample():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showuse", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        pick = lambda pat: next((v for k, v in card.items() if re.search(pat, k, re.I)), None)
        return {"power_w": pick(r"average.*power|socket.*power|current.*power"), "sclk": pick(r"sclk clock speed"), "mclk": pick(r"mclk clock speed"),
                "busy": pick(r"GPU use")}
    except Exception as e:                                   # noqa: BLE001
        return {"error": repr(e)[:200]}


def main():
    import torch
    from omniparser_amd import _lib as L
    from omniparser_amd.planner import PlanBuilder, View
    dev = "cuda"
    stream = torch.cuda.Stream()

    def gemm_plan(M, N, K, act, res):
        pb = PlanBuilder(dev, L.F32)
        x = View(torch.randn(1, M, 1, K, device=dev), 0, K)
        pre = PlanBuilder(dev, L.F32); pre.split_convert(x); pre.build().run(stream); stream.synchronize()
        x.fmt = "split"
        w = pb.pack_weight_dma(torch.randn(N, K) * 0.05)
        y = pb.alloc(1, M, 1, N)
        r = View(torch.randn(1, M, 1, N, device=dev), 0, N) if res else None
        pb.conv(x, w, torch.randn(N), y, 1, act=act, res=r, out_split=(act == L.ACT_GELU))
        p = pb.build(); p._x = (x, r)
        return p, 2.0 * M * N * K

    def ln_plan(rows, C):
        pb = PlanBuilder(dev, L.F32)
        x = torch.randn(rows, C, device=dev); y = torch.empty_like(x)
        g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        pb.add_op(L.make_op(L.OP_LAYERNORM, L.F32, p=[x.data_ptr(), None, g.data_ptr(), b.data_ptr(), y.data_ptr(), None], i={0: rows, 1: 1, 3: C}, f={0: 1e-5}))
        p = pb.build(); p._x = (x, y, g, b)
        return p, 8.0 * rows * C

    cases = [("gemm_fc2_K2048", *gemm_plan(294912, 512, 2048, L.ACT_NONE, True), "TFLOP/s", 1e12),
             ("gemm_fc1_gelu_K512", *gemm_plan(294912, 2048, 512, L.ACT_GELU, False), "TFLOP/s", 1e12),
             ("layernorm_4.7Mx128", *ln_plan(4718592, 128), "TB/s", 1e12)]
    out = {"idle": [smi_sample() for _ in range(3)], "cases": {}}
    for name, plan, work, unit, scale in cases:
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append(smi_sample())
                time.sleep(0.25)
        for _ in range(5):
            plan.run(stream)
        stream.synchronize()
        th = threading.Thread(target=poll); th.start()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 6.0:
            for _ in range(20):
                plan.run(stream)
            stream.synchronize(); n += 20
        sec = time.perf_counter() - t0
        stop.set(); th.join()
        out["cases"][name] = {"launches": n, "rate": round(work * n / sec / scale, 2), "unit": unit, "ms_per_launch": round(1000 * sec / n, 4), "samples": samples}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
