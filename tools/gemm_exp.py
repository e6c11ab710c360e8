"""The experiment round 5 specified and did not run (profiles/r5_gemm_bound_analysis.md): what bounds
`gemm_dma_kernel` on the MI355X — the operand FEED (64 KB of LDS-DMA per 32-wide K slice and CU, ~7 TB/s out of L2 across the chip) or
the POWER the three f16 MFMA products per MAC draw (board limit 1400 W, clock pulled to 1.45-1.77 GHz)?

Variants of the SAME kernel (256x256 tile, 8 waves, 2-stage ring), selected at COMPILE time by -DOMNI_GEMM_EXP=<bits> in a patched COPY
of csrc/gemm_dma.hip — the shipping library is not touched and carries no knob:

  0   shipping kernel (3 products per MAC, full DMA)
  1   2 products (w_lo * x_hi dropped)                    = round 5's ablation (-11 % time at -33 % MFMAs)
  3   2 products + hi halves only DMA'd                   = the separating experiment: same MFMA count as 1, HALF the LDS-DMA bytes
  2   3 products + hi halves only DMA'd                   = the feed halved at the full MFMA count
  4   no MFMA (DMA + ds_read + barriers + epilogue only)  = what the feed costs alone
  8   MFMA only (no DMA, no ds_read inside the K loop)    = what the matrix pipes cost alone (fragments loaded once per tile)

(hi-only DMA: the lanes of an LDS-DMA instruction whose source chunk holds lo halves are masked off after the first slice of a tile has
filled both halves, so the lo products keep reading real — stale — lo data and the operand toggling stays comparable.)

  python tools/gemm_exp.py build                 CPU container: .exp/libomni_amd_exp<bits>.so for every variant (travels with gpurun)
  python tools/gemm_exp.py run [seconds]         GPU box: every variant x {fc2 K = 2048, fc1 GELU K = 512} back to back under rocm-smi
                                                 sampling (power, sclk), one subprocess per variant -> JSON lines on stdout
Results: profiles/r6_gemm_exp.json (+ the reading in DESIGN.md section 3e).  Numbers of variants 1-8 are WRONG on purpose."""
import json
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
EXP = ROOT / ".exp"
VARIANTS = (0, 1, 3, 2, 4, 8)


def patched_source() -> str:
    s = (ROOT / "omniparser_amd" / "csrc" / "gemm_dma.hip").read_text()

    def rep(old, new):
        nonlocal s
        assert s.count(old) == 1, (s.count(old), old[:80])
        s = s.replace(old, new)

    rep('#include "gemm_common.h"', '#include "gemm_common.h"\n#ifndef OMNI_GEMM_EXP\n#define OMNI_GEMM_EXP 0\n#endif')
    # ---- products
    rep('''  auto mma = [&](const AF& af, const WF& wf, int ip) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[ip * 2 + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h[j], af.h[t], acc[ip * 2 + t][j], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[ip * 2 + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.l[j], af.h[t], acc[ip * 2 + t][j], 0, 0, 0);
''', '''  auto mma = [&](const AF& af, const WF& wf, int ip) {
#if (OMNI_GEMM_EXP & 4)
#pragma unroll
    for (int t = 0; t < 2; ++t) asm volatile("" ::"v"(af.h[t]), "v"(af.l[t]));
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(wf.h[j]), "v"(wf.l[j]));
    return;
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[ip * 2 + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h[j], af.h[t], acc[ip * 2 + t][j], 0, 0, 0);
#if !(OMNI_GEMM_EXP & 1)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[ip * 2 + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.l[j], af.h[t], acc[ip * 2 + t][j], 0, 0, 0);
#endif
''')
    rep("  constexpr int NM = 6 * TN;  ", "  constexpr int NM = ((OMNI_GEMM_EXP & 4) ? 0 : (OMNI_GEMM_EXP & 1) ? 4 : 6) * TN;  ")
    # ---- hi-only DMA: lanes whose SOURCE chunk (slot ^ swizzle) holds lo halves (chunk & 2) are masked off for slices >= NSTAGE
    rep('''  auto issue_piece = [&](int kt, int stage, int i) {
    const int so = kt * 128;''', '''  auto issue_piece = [&](int kt, int stage, int i) {
    const int so = kt * 128;
#if (OMNI_GEMM_EXP & 8)
    if (kt >= NSTAGE - 1) return;                    // MFMA only: the ring is filled once per tile
#endif
#if (OMNI_GEMM_EXP & 2)
    {
      const int rl8 = ((i < A_DMA ? (wave * A_DMA + i) : (wave * B_DMA + i - A_DMA)) * 8 + rsub);
      if (kt >= NSTAGE && (((slot ^ ((rl8 >> 1) & 7)) & 2) != 0)) return;      // exec-masked: this lane's 16 bytes are lo halves
    }
#endif''')
    # ---- MFMA only: fragments are read in the first slice of a tile, afterwards the registers are reused
    rep('''    const unsigned char* st = lds + stage * STAGE;
    loadW(st, 0, wf[0]);
    loadA(st, 0, 0, af[0]);''', '''    const unsigned char* st = lds + stage * STAGE;
#if (OMNI_GEMM_EXP & 8)
    if (kt == 0) {
      loadW(st, 0, wf[0]); loadW(st, 1, wf[1]);
      loadA(st, 0, 0, af[0]); loadA(st, 1, 0, af[1]);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) { mma(af[u & 1], wf[(u / NP) & 1], u % NP); __builtin_amdgcn_sched_barrier(0); }
    stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    nstage = nstage + 1 == NSTAGE ? 0 : nstage + 1;
    return;
#endif
    loadW(st, 0, wf[0]);
    loadA(st, 0, 0, af[0]);''')
    return s


def build():
    from omniparser_amd.build import FLAGS, HIPCC, OBJ, build_lib, sources
    build_lib(verbose=False)                                   # the shipping objects (everything but gemm_dma.hip is linked as is)
    EXP.mkdir(exist_ok=True)
    csrc = ROOT / "omniparser_amd" / "csrc"
    src = EXP / "gemm_dma_exp.hip"
    src.write_text(patched_source())
    for v in VARIANTS:
        obj = EXP / f"gemm_dma_exp{v}.o"
        subprocess.run([HIPCC, *FLAGS, f"-DOMNI_GEMM_EXP={v}", f"-I{csrc}", "-c", str(src), "-o", str(obj)], check=True)
        objs = [str(obj) if s.stem == "gemm_dma" else str(OBJ / (s.stem + ".o")) for s in sources()]
        lib = EXP / f"libomni_amd_exp{v}.so"
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *objs], check=True)
        print("built", lib, flush=True)
    # the record of what was compiled: a unified diff against the shipping source
    d = subprocess.run(["diff", "-u", str(csrc / "gemm_dma.hip"), str(src)], capture_output=True, text=True).stdout
    (ROOT / "tools" / "archive" / "r6_gemm_exp.patch").write_text(d)


def run_variant(v, seconds: float):
    import torch
    from omniparser_amd import _lib as L
    from tools.power_trace import smi_sample
    if str(v).startswith("s"):                    # "s1" / "s2": the SHIPPING library with OMNI_GEMM_SCHED unset / = 2 (ping-pong schedule A/B)
        os.environ["OMNI_GEMM_SCHED"] = v[1:]
        L.lib()
    else:
        L._lib = L.bind(EXP / f"libomni_amd_exp{v}.so")
    from omniparser_amd.planner import PlanBuilder, View
    dev = "cuda"
    stream = torch.cuda.Stream()

    def gemm_plan(M, N, K, act, res):
        pb = PlanBuilder(dev, L.F32)
        g = torch.Generator(device="cpu").manual_seed(0)
        x = View(torch.randn(1, M, 1, K, generator=g).to(dev), 0, K)
        pre = PlanBuilder(dev, L.F32); pre.split_convert(x); pre.build().run(stream); stream.synchronize()
        x.fmt = "split"
        w = pb.pack_weight_dma(torch.randn(N, K, generator=g) * 0.05)
        y = pb.alloc(1, M, 1, N)
        r = View(torch.randn(1, M, 1, N, generator=g).to(dev), 0, N) if res else None
        pb.conv(x, w, torch.randn(N, generator=g), y, 1, act=act, res=r, out_split=(act == L.ACT_GELU))
        p = pb.build(); p._x = (x, r)
        return p, 2.0 * M * N * K

    out = {"variant": v, "idle": smi_sample(), "cases": {}}
    for name, (M, N, K, act, res) in (("fc2_K2048_N512_res", (294912, 512, 2048, L.ACT_NONE, True)),
                                      ("fc1_K512_N2048_gelu_split", (294912, 2048, 512, L.ACT_GELU, False)),
                                      ("qkv_K512_N1536", (294912, 1536, 512, L.ACT_NONE, False))):
        plan, work = gemm_plan(M, N, K, act, res)
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append(smi_sample())
                time.sleep(0.2)
        for _ in range(5):
            plan.run(stream)
        stream.synchronize()
        # HIP-event time of 20 back-to-back launches first (no sampler thread), then the sampled run
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(20):
                plan.run(stream)
            e1.record(stream)
        stream.synchronize()
        ms20 = e0.elapsed_time(e1) / 20
        th = threading.Thread(target=poll); th.start()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                plan.run(stream)
            stream.synchronize(); n += 20
        sec = time.perf_counter() - t0
        stop.set(); th.join()

        def num(s):
            try:
                return float(str(s).strip("()MHzWw% "))
            except (TypeError, ValueError):
                return None
        pw = [num(s.get("power_w")) for s in samples[2:] if num(s.get("power_w")) is not None]
        ck = [num(s.get("sclk")) for s in samples[2:] if num(s.get("sclk")) is not None]
        out["cases"][name] = {"ms_per_launch_events_20": round(ms20, 4), "ms_per_launch_sustained": round(1000 * sec / n, 4),
                              "algorithmic_tflops_sustained": round(work * n / sec / 1e12, 1), "launches": n,
                              "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": max(pw) if pw else None,
                              "sclk_mhz_mean": round(sum(ck) / len(ck), 1) if ck else None, "samples": len(samples)}
        del plan
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "build"
    if mode == "build":
        build()
    elif mode == "variant":
        run_variant(sys.argv[2] if sys.argv[2].startswith("s") else int(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 4.0)
    elif mode == "sched":                          # python tools/gemm_exp.py sched [seconds]: shipping vs ping-pong, interleaved twice
        secs = sys.argv[2] if len(sys.argv) > 2 else "4"
        for v in ("s1", "s2", "s3", "s1", "s2", "s3"):
            r = subprocess.run([sys.executable, __file__, "variant", v, secs], capture_output=True, text=True, timeout=600)
            line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
            print(line or json.dumps({"variant": v, "error": (r.stderr or r.stdout)[-600:]}), flush=True)
    else:
        secs = sys.argv[2] if len(sys.argv) > 2 else "4"
        for v in VARIANTS:
            r = subprocess.run([sys.executable, __file__, "variant", str(v), secs], capture_output=True, text=True, timeout=600)
            line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
            print(line or json.dumps({"variant": v, "error": (r.stderr or r.stdout)[-600:]}), flush=True)
