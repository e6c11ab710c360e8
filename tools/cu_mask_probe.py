"""Does partitioning the chip pay?  (hardware probe for round 4, written without a GPU at the end of round 3)

The caption encode is a chain of MFMA-bound GEMMs (480 ms of a 709 ms step; power-bound: matrix pipe 60 % busy at a 1.73 GHz
sustained clock) and HBM-bound kernels (220 ms: depthwise conv + LayerNorm, attention, channel attention) that run at 2.1 GHz.  The
two encode lanes of `parse_stream` overlap them only marginally (+2-3 %): a GEMM launch owns every CU's registers and LDS, so the
other lane's kernel waits.  With CU-masked streams (hipExtStreamCreateWithCUMask, `L.masked_stream`) the lanes get disjoint CU sets
and really run side by side — worth it if a power-bound GEMM loses little on fewer CUs.  This tool measures exactly that:
  1. one GEMM (DaViT stage-2 fc2 shape) and one HBM-bound op (LayerNorm over the same tokens) under masks of 256 / 192 / 128 / 64
     CUs in several layouts (low / high / interleaved halves, 32-CU words) -> how the runtime maps mask bits, how each kernel scales;
  2. eager launches vs a captured plan replayed on a masked stream (do hipGraph launches honour the stream's mask?);
  3. GEMMs on one stream and LayerNorms on another, both unmasked vs disjoint masks: wall time against the serial sum.
usage (GPU box): python tools/cu_mask_probe.py > gpurun_out/r4/cu_mask_probe.json   (~1 GPU-minute)"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

M, K, N = 147456, 2048, 512          # half a 128-crop stage-2 fc2 launch: 0.93 TFLOP x 3 products


def parse_mask(spec, total=256):
    from omniparser_amd._lib import parse_cu_spec
    return parse_cu_spec(spec, total)


# Measured at the end of round 3 (profiles/r3_cu_mask_probe.md): bit j = CU j / 8 of XCD j % 8, and an XCD whose share of the mask is
# empty is NOT restricted — strided sets ("0-255:2") are no-ops; a partition gives every XCD a share: contiguous ranges, multiples of 8.
MASKS = {
    "all256": "0-255", "low192": "0-191", "low160": "0-159", "low128": "0-127", "high128": "128-255", "high96": "160-255",
    "low64": "0-63", "high64": "192-255", "word0": "0-31", "even128_noop": "0-255:2",
}


def build_ops(torch, L, PlanBuilder, View, dev):
    """(gemm plan, layernorm plan): each one op, operands resident."""
    pb = PlanBuilder(dev, L.F32)
    x = View((torch.randn(1, M, 1, K, device=dev) * 0.5), 0, K)
    pre = PlanBuilder(dev, L.F32)
    pre.split_convert(x)
    pre.build().run(None)
    torch.cuda.synchronize()
    x.fmt = "split"
    y = pb.alloc(1, M, 1, N)
    r = View(torch.randn(1, M, 1, N, device=dev), 0, N)
    w = pb.pack_weight_dma(torch.randn(N, K) * 0.05)
    pb.conv(x, w, torch.randn(N), y, 1, res=r)
    gemm = pb.build()
    pl = PlanBuilder(dev, L.F32)
    t = pl.alloc(1, M, 1, N)
    t.t.normal_()
    o = pl.alloc(1, M, 1, N)
    g, b = pl.upload(torch.ones(N)), pl.upload(torch.zeros(N))
    pl.add_op(L.make_op(L.OP_LAYERNORM, L.F32, p=[t.ptr, None, g.data_ptr(), b.data_ptr(), o.ptr, None],
                        i={0: M, 1: 1, 3: N, 5: 0, 6: 0}, f={0: 1e-5}))
    ln = pl.build()
    # the pipeline's heaviest HBM-bound kernel: depthwise conv + LayerNorm, C = 512 on 48x48 tokens (214 registers: it cannot
    # co-reside with gemm_dma_kernel blocks the way the light LayerNorm does)
    pd = PlanBuilder(dev, L.F32)
    Bd, Hd, Cd = M // (48 * 48), 48, 512
    xd, y1, hd = pd.alloc(Bd, Hd, Hd, Cd), pd.alloc(Bd, Hd, Hd, Cd), pd.alloc(Bd, Hd, Hd, Cd)
    xd.t.normal_()
    wd, bd = pd.upload(torch.randn(3, 3, Cd) * 0.1), pd.upload(torch.zeros(Cd))
    gd, bb = pd.upload(torch.ones(Cd)), pd.upload(torch.zeros(Cd))
    pd.add_op(L.make_op(L.OP_DWCONV3_LN, L.F32, p=[xd.ptr, wd.data_ptr(), bd.data_ptr(), hd.ptr, y1.ptr, gd.data_ptr(), bb.data_ptr()],
                        i={0: Bd, 1: Hd, 2: Hd, 3: Cd, 6: 0}, f={0: 1e-5}))
    build_ops.dwln = (pd.build(), 3.0 * 4 * Bd * Hd * Hd * Cd)
    return gemm, ln, 2.0 * M * N * K, 2.0 * 4 * M * N


def timed(torch, stream, fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()                                         # warm
    stream.synchronize()
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    stream.synchronize()
    return max(e0.elapsed_time(e1) / iters, 1e-6)


def main(make_stream=None, iters=6):
    import torch
    from omniparser_amd import _lib as L
    from omniparser_amd.planner import PlanBuilder, View
    dev = L.require_device("cuda", "cu_mask_probe")
    make_stream = make_stream or (lambda spec: L.masked_stream(dev, L.cu_mask_words(parse_mask(spec))))
    gemm, ln, flops, ln_bytes = build_ops(torch, L, PlanBuilder, View, dev)
    dwln, dwln_bytes = build_ops.dwln
    out = {"shape": {"M": M, "K": K, "N": N}, "single": {}, "graph": {}, "concurrent": {}}
    streams = {}
    for name, spec in MASKS.items():
        try:
            st = streams[name] = make_stream(spec)
            g_ms = timed(torch, st, lambda: gemm.run(st), iters)
            l_ms = timed(torch, st, lambda: ln.run(st), iters)
            d_ms = timed(torch, st, lambda: dwln.run(st), iters)
            out["single"][name] = {"cus": len(parse_mask(spec)), "gemm_ms": round(g_ms, 4), "gemm_tflops": round(flops / g_ms / 1e9, 1),
                                   "layernorm_ms": round(l_ms, 4), "layernorm_tbps": round(ln_bytes / l_ms / 1e9, 2),
                                   "dwconv_ln_ms": round(d_ms, 4), "dwconv_ln_tbps": round(dwln_bytes / d_ms / 1e9, 2)}
        except Exception as e:                                     # noqa: BLE001 — report, keep going
            out["single"][name] = {"error": repr(e)[:300]}
    # 2. does a captured plan replayed on a masked stream stay inside the mask?  (same time as the eager masked launch = yes)
    for name in ("all256", "low64"):
        try:
            st = streams[name]
            gemm.capture(st)
            st.synchronize()
            out["graph"][name] = {"replay_ms": round(timed(torch, st, lambda: gemm.replay(st), iters), 4),
                                  "eager_ms": out["single"][name]["gemm_ms"]}
        except Exception as e:                                     # noqa: BLE001
            out["graph"][name] = {"error": repr(e)[:300]}
    # 3. GEMMs on A, LayerNorms on B: wall time of both queues together against the serial sum
    n_g, n_l = 4, 12
    for a, b in (("all256", "all256"), ("low192", "high64"), ("low160", "high96"), ("low128", "high128")):
        try:
            sa = streams[a]
            sb = make_stream(MASKS[b]) if a == b else streams[b]
            for s_ in (sa, sb):
                s_.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_g):
                gemm.run(sa)
            for _ in range(n_l):
                ln.run(sb)
            sa.synchronize(); sb.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            serial = n_g * out["single"]["all256"]["gemm_ms"] + n_l * out["single"]["all256"]["layernorm_ms"]
            out["concurrent"][f"{a}|{b}"] = {"wall_ms": round(wall, 3), "serial_unmasked_ms": round(serial, 3), "ratio": round(wall / max(serial, 1e-9), 3)}
        except Exception as e:                                     # noqa: BLE001
            out["concurrent"][f"{a}|{b}"] = {"error": repr(e)[:300]}
    # 4. the pipeline's shape: two lanes running the SAME mixed queue (GEMM, 2 x depthwise conv + LayerNorm) x 6, the second lane half a period ahead,
    # unmasked against symmetric halves — wall time per lane-iteration
    out["two_mixed_lanes"] = {}
    for a, b in (("all256", "all256"), ("low128", "high128")):
        try:
            sa = streams[a]
            sb = make_stream(MASKS[b]) if a == b else streams[b]
            for s_ in (sa, sb):
                s_.synchronize()
            t0 = time.perf_counter()
            for it in range(6):
                gemm.run(sa)
                for _ in range(2):
                    dwln.run(sb)
                for _ in range(2):
                    dwln.run(sa)
                gemm.run(sb)
            sa.synchronize(); sb.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            serial = 12 * out["single"]["all256"]["gemm_ms"] + 24 * out["single"]["all256"]["dwconv_ln_ms"]
            out["two_mixed_lanes"][f"{a}|{b}"] = {"wall_ms": round(wall, 3), "serial_unmasked_ms": round(serial, 3),
                                                   "ratio": round(wall / max(serial, 1e-9), 3)}
        except Exception as e:                                     # noqa: BLE001
            out["two_mixed_lanes"][f"{a}|{b}"] = {"error": repr(e)[:300]}
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main()
