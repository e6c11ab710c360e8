"""The HIP kernels run FROM THEIR DEVICE SOURCES on the host emulation (tests/emu: csrc/*.hip compiled for the host, work-items as
fibers, waves as collectives, MFMA / LDS-DMA / barriers emulated) through the same C ABI and the same checks as the `-m gpu` tests
(tests/gpu_checks.py), at sizes a CPU finishes in seconds.  What this covers: every kernel's algorithm, indexing, barrier and
`s_waitcnt` placement, LDS layouts, the launch geometry computed by the host side of the library.  What it cannot cover: anything
only the hardware decides (timing, LDS capacity, hipGraph replay) — the `-m gpu` tests are the parity statement."""
import pytest

from omniparser_amd import _lib as L

SMALL_CONV = [
    # B, H, W, Cin, Cout, k, s, in_ld, in_off, out_ld, out_off, res, act
    (1, 20, 24, 64, 64, 1, 1, 64, 0, 64, 0, False, L.ACT_SILU),
    (2, 17, 23, 32, 96, 3, 1, 64, 32, 128, 32, True, L.ACT_SILU),
    (1, 24, 24, 128, 256, 3, 2, 128, 0, 256, 0, False, L.ACT_SILU),
    (1, 33, 31, 8, 40, 3, 2, 8, 0, 40, 0, False, L.ACT_NONE),          # generic (unaligned Cin) path
    (1, 12, 12, 1024, 256, 1, 1, 1024, 0, 256, 0, False, L.ACT_GELU),  # small M, large K (split-K)
    (1, 9, 9, 256, 1, 1, 1, 256, 0, 1, 0, False, L.ACT_NONE),          # Cout = 1 (class head)
    (3, 12, 12, 64, 64, 3, 1, 192, 64, 64, 0, True, L.ACT_NONE),
]
SMALL_GEMM_DMA = [
    # M, K, N, in_ld, in_off, out_ld, out_off, res, act, out_split
    (300, 128, 128, 128, 0, 128, 0, False, L.ACT_NONE, False),
    (333, 256, 384, 256, 0, 384, 0, True, L.ACT_NONE, False),
    (260, 128, 512, 128, 0, 512, 0, False, L.ACT_GELU, True),
    (513, 32, 256, 64, 32, 512, 256, False, L.ACT_NONE, False),
    (5, 512, 256, 512, 0, 256, 0, False, L.ACT_NONE, False),
    (70, 64, 128, 64, 0, 128, 0, False, L.ACT_NONE, False),        # two K slices: prologue + drain only
    (70, 96, 256, 96, 0, 256, 0, True, L.ACT_NONE, False),          # three K slices
]


def test_mfma_fragment_layout(emu):
    import gpu_checks as G
    G.check_mfma_layout()


@pytest.mark.parametrize("dtype", [L.F32, L.F16])
def test_conv_igemm(emu, dtype):
    import gpu_checks as G
    r = G.check_conv(dtype, cases=SMALL_CONV)
    assert r["cases"] >= 6
    if dtype == L.F32:      # split-f16 convs: in-launch split-K combine == reduce launch, bit for bit (blocks run on concurrent host threads)
        assert r["splitk_combine_vs_reduce_bitwise"] >= 12, r


def test_conv_patch_rows(emu):
    """row-patch form of the first patch embedding (OMNI_OP_CONV i25) vs f64 and vs the exact-f32 kernel, borders included"""
    import gpu_checks as G
    r = G.check_conv_patch()
    assert r["cases"] == 6, r


def test_conv_igemm_exact_f32_path(emu, monkeypatch):
    import gpu_checks as G
    monkeypatch.setenv("OMNI_CONV_SPLIT", "0")
    r = G.check_conv(L.F32, cases=SMALL_CONV)
    assert r["cases"] >= 6 and r["worst_rel_err"] < 2e-5


def test_gemm_dma_presplit_every_tile(emu):
    import gpu_checks as G
    r = G.check_gemm_dma(cases=SMALL_GEMM_DMA)
    assert r["cases"] == 3 * len(SMALL_GEMM_DMA) and r["worst_rel_err"] < 2e-6


def test_gemm_ping_pong_schedule_equals_the_lockstep_schedule_bit_for_bit(emu):
    """the 256x256 tile's ping-pong K loop (round 6: the two M halves half a phase apart, LDS-DMA pieces staged by token half with counted
    vmcnt) against the lockstep loop on the host emulation — where an LDS-DMA piece lands at the issuer's covering vmcnt and not before,
    so a fragment read placed ahead of its wait + barrier reads stale bytes: 1, 2, 3, 5 and 16 K slices, ragged M, every epilogue."""
    import gpu_checks as G
    out = G.check_gemm_schedules_bitwise(M=700, cases=((96, 256, False, L.ACT_NONE, True), (32, 512, True, L.ACT_GELU, False),
                                                       (64, 256, False, L.ACT_NONE, False), (160, 256, True, L.ACT_NONE, False),
                                                       (512, 256, False, L.ACT_NONE, True)))
    assert len(out) == 5, out


def test_split_range_guard_counts_values_beyond_f16(emu):
    """the split formats clamp / lose |x| > 65504 and the fp32 reference does not — the producing kernels count it
    (omni_overflow_count), the pipeline surfaces it in stats["split_overflow"] and raises under OMNI_STRICT_RANGE=1."""
    import gpu_checks as G
    out = G.check_range_guard()
    assert out["healthy"] == 0 and out["gemm_split_output"] >= 1


def test_greedy_step_never_emits_an_out_of_range_id(emu):
    import gpu_checks as G
    G.check_greedy_degenerate_rows()


def test_mlp_fused_vs_f64_and_vs_two_launches(emu):
    import gpu_checks as G
    r = G.check_mlp_fused(cases=((140, 128, 0, 128, 0), (65, 192, 48, 160, 16)))
    assert r["worst_rel_err"] < 3e-6


@pytest.mark.parametrize("dtype", [L.F32, L.F16])
def test_pool_resize_letterbox(emu, dtype):
    import gpu_checks as G
    G.check_pools(dtype)
    G.check_letterbox(dtype, sizes=((320, 200, 160), (301, 199, 160), (160, 120, (120, 160))))


@pytest.mark.parametrize("seed,nc,frac", [(0, 1, 0.08), (1, 3, 0.05), (2, 1, 0.6), (3, 2, 0.002)])
def test_decode_nms_vs_oracle(emu, seed, nc, frac):
    import gpu_checks as G
    r = G.check_post(seed=seed, nc=nc, frac=frac)
    assert r["nms_exact_on_gpu_candidates"] and (r["min_iou"] is None or r["min_iou"] >= 0.999)


def test_nms_known_answers(emu):
    import gpu_checks as G
    G.check_nms_known_answers()


@pytest.mark.parametrize("dtype", [L.F32, L.F16])
def test_caption_kernels_vs_interpreter(emu, dtype):
    import gpu_checks as G
    r = G.check_caption_ops(dtype)
    assert dtype != L.F32 or ("attn_window_13" in r and "chan_attn_300" in r and "attn_plain_200_sharp" in r)


def test_presplit_gemm_accuracy_versus_activation_scale(emu):
    """Format B stores an activation as hi + lo with BOTH halves f16: once |x| < 2^-4 the lo half is subnormal and the pair resolves
    x to 2^-24 absolute, i.e. 2^-24 / |x| relative.  f32-class accuracy therefore holds for O(1) activations — LayerNorm outputs,
    which feed most of these GEMMs, are unit-variance by construction — and degrades gracefully (never worse than f16) below.  This
    pins the measured error of the kernel against that model; DESIGN.md section 4 states the limit."""
    import math
    import torch
    import gpu_checks as G
    from plan_interp import split_decode
    from omniparser_amd.planner import PlanBuilder, View
    g = torch.Generator().manual_seed(3)
    M, K, N = 300, 256, 128
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    x0 = torch.randn(M, K, generator=g)
    seen = {}
    for e in (0, -4, -8, -12):
        x = x0 * 2.0 ** e
        pb = PlanBuilder("cpu", L.F32)
        xv = View(x.clone().view(1, M, 1, K), 0, K)
        pb.split_convert(xv)
        ov = View(torch.zeros(1, M, 1, N), 0, N)
        pb.conv(xv, pb.pack_weight_dma(w), None, ov, 1)
        for op in pb.ops:
            L.launch(op)
        ref = x.double() @ w.double().t()
        err = G.rel_err(ov.t.view(M, N).double(), ref)
        # against the operands as the kernel sees them the product itself stays exact to f32 class at every scale
        seen_x = split_decode(xv.t.view(M, K).contiguous()).double()
        kern = G.rel_err(ov.t.view(M, N).double(), seen_x @ w.double().t())
        seen[e] = (err, kern)
        assert kern < 2e-6, (e, kern)
        assert err < max(2e-6, 4 * 2.0 ** (-24 - e)), (e, err)       # representation limit: 2^-24 absolute per element
    assert seen[0][0] < 1e-6 and seen[-12][0] > seen[0][0]
    print(seen)


def test_kernel_checks_also_pass_in_reverse_work_item_order():
    """The emulation runs the work-items of a workgroup in ascending order between synchronisation points; OMNI_EMU_ORDER=reverse runs
    them in descending order.  A missing __syncthreads / s_waitcnt between "write my slot" and "read a neighbour's" corrupts the
    result under at least one of the two orders — so the kernel checks above (and the hand-off / overlay / PNG ones) must pass under
    both.  (Own process: the order is read once per process.)"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    here = Path(__file__).resolve().parent
    env = dict(os.environ, OMNI_EMU_ORDER="reverse")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-k", "not reverse_work_item_order",
                        str(here / "test_kernels_emu_cpu.py"), str(here / "test_glue_emu_cpu.py"), str(here / "test_overlay_png_emu_cpu.py")],
                       env=env, capture_output=True, text=True, cwd=str(here.parent))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_dwconv3_strip_and_point_kernels_equal_torch(emu):
    """x + depthwise_conv3x3(x) + bias from the device sources: the strip kernel serves power-of-two vector counts per pixel (every
    DaViT stage), the point kernel the rest; shapes cover H % 4 != 0, one-row / one-column images, rows shorter than a block, several
    row chunks per row, f32 and f16.  (Until round 4 a debug entry point of the SHIPPING library ran these bodies on the host; the
    emulation build runs the kernels themselves.)"""
    import torch
    import torch.nn.functional as Fn
    g = torch.Generator().manual_seed(11)
    for dtype, tdt, V, tol in ((L.F32, torch.float32, 4, 2e-5), (L.F16, torch.float16, 8, 1e-2)):
        for (B, H, W, C) in [(2, 12, 12, 128), (1, 7, 5, 256), (3, 1, 9, 64), (1, 9, 1, 32), (2, 6, 50, 64), (1, 5, 3, 1024), (1, 4, 4, 96),
                             (1, 13, 11, 8 * V // 4)]:
            x = torch.randn(B, H, W, C, generator=g).to(tdt).contiguous()
            w = (torch.randn(3, 3, C, generator=g) * 0.3).to(tdt).contiguous()
            bias = torch.randn(C, generator=g).contiguous()
            y = torch.full_like(x, float("nan"))
            op = L.make_op(L.OP_DWCONV3, dtype, p=[x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, y.data_ptr()], i={0: B, 1: H, 2: W, 3: C})
            L.launch(op, 0)
            assert not torch.isnan(y.float()).any(), (B, H, W, C)                   # every output written
            ref = x.float() + Fn.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(2, 0, 1).unsqueeze(1), bias, padding=1,
                                        groups=C).permute(0, 2, 3, 1)
            assert (y.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), (B, H, W, C, dtype)
