"""`-m gpu` parity tests of the HIP kernels, driven through the C ABI (see tests/gpu_checks.py)."""
import pytest

from omniparser_amd import _lib as L

pytestmark = pytest.mark.gpu


def test_first_launch_canary():
    """the first device launch of the library in this process, step by step (round 3's driver run died here without a trail)"""
    import gpu_checks as G
    trail = G.check_first_launch_canary()
    assert trail["launch_null_stream"] and trail["launch_explicit_stream"]


def test_bad_pointers_are_errors_not_faults():
    import gpu_checks as G
    seen = G.check_bad_pointers_are_errors()
    if "skipped" in seen:
        import pytest
        pytest.skip(seen["skipped"])
    assert len(seen) == 8, seen
    assert "not inside any device allocation" in seen["wild/launch"] and "runs past its allocation" in seen["past_end/plan"], seen


def test_split_range_guard_counts_values_beyond_f16():
    """|x| > 65504 cannot be held by the split-f16 GEMM operands (the fp32 reference has no such limit, ref:util/utils.py:66): every
    producing kernel counts it, `omni_overflow_count` reads it; 0 on healthy tensors."""
    import gpu_checks as G
    out = G.check_range_guard()
    assert out["healthy"] == 0 and out["split_convert_input_7e4"] >= 1 and out["gemm_split_output"] >= 1 and out["conv_f32_output"] >= 1, out


def test_gemm_k_loop_schedules_are_bit_identical():
    """lockstep vs ping-pong K-loop schedule of the 256x256 LDS-DMA GEMM tile: same bits on 70 000 rows, five shapes."""
    import gpu_checks as G
    out = G.check_gemm_schedules_bitwise()
    assert len(out) == 5, out


def test_mfma_fragment_layout():
    import gpu_checks as G
    G.check_mfma_layout()


@pytest.mark.parametrize("dtype", [L.F32, L.F16])
def test_conv_igemm(dtype):
    """default paths: split-f16 MFMA for long-K f32 GEMMs, exact-f32 / f16 MFMA otherwise."""
    import gpu_checks as G
    r = G.check_conv(dtype)
    assert r["cases"] >= 6
    if dtype == L.F32:      # round 6: the in-launch split-K combine equals the reduce launch bit for bit, counters self-reset
        assert r["splitk_combine_vs_reduce_bitwise"] >= 24, r


def test_conv_patch_rows():
    """round 6: the captioner's 7 x 7 / stride-4 patch embedding over 4 stored channels as a 7 x 1 convolution over 8 consecutive pixels on
    the split-f16 MFMA kernel (OMNI_OP_CONV i25) vs an f64 convolution and vs the exact-f32 kernel it replaces, image borders included."""
    import gpu_checks as G
    r = G.check_conv_patch()
    print(r)
    assert r["cases"] == 6, r


def test_conv_igemm_exact_f32_path(monkeypatch):
    """OMNI_CONV_SPLIT=0: every f32 GEMM on v_mfma_f32_32x32x2_f32."""
    import gpu_checks as G
    monkeypatch.setenv("OMNI_CONV_SPLIT", "0")
    r = G.check_conv(L.F32)
    assert r["cases"] >= 6 and r["worst_rel_err"] < 2e-5


def test_gemm_dma_presplit():
    """pre-split LDS-DMA GEMM (every tile configuration), split_convert and LayerNorm's format-B outputs."""
    import gpu_checks as G
    r = G.check_gemm_dma()
    assert r["cases"] >= 20 and r["worst_rel_err"] < 2e-6


def test_mlp_fused():
    """OMNI_OP_MLP_FUSED (DaViT stage-0 FFN in one kernel, hidden activations in registers) vs f64 and vs fc1 + fc2 as two launches;
    then at the benched row count (36 864 tokens x 8 crops) against the two-launch composition only."""
    import gpu_checks as G
    r = G.check_mlp_fused()
    assert r["cases"] >= 4 and r["worst_rel_err"] < 3e-6
    r = G.check_mlp_fused(seed=1, cases=((8 * 36864, 128, 0, 128, 0),))
    assert r["worst_vs_two_launches"] < 3e-6


def test_greedy_step_degenerate_rows():
    """all-NaN / all -inf logit rows (padding rows fed recycled memory, a diverged input) yield token 0, never an out-of-range id."""
    import gpu_checks as G
    G.check_greedy_degenerate_rows()


@pytest.mark.parametrize("dtype", [L.F32, L.F16])
def test_pool_and_resize(dtype):
    import gpu_checks as G
    G.check_pools(dtype)


@pytest.mark.parametrize("dtype", [L.F32, L.F16])
def test_letterbox_matches_pillow(dtype):
    import gpu_checks as G
    G.check_letterbox(dtype)


@pytest.mark.parametrize("seed,nc,frac", [(0, 1, 0.08), (1, 3, 0.05), (2, 1, 0.6), (3, 2, 0.002)])
def test_decode_nms_vs_oracle(seed, nc, frac):
    import gpu_checks as G
    r = G.check_post(seed=seed, nc=nc, frac=frac)
    assert r["nms_exact_on_gpu_candidates"] and (r["min_iou"] is None or r["min_iou"] >= 0.999)


def test_decode_nms_native_size():
    """A = 42 840 anchors (1088x1920 network input), thousands of candidates."""
    import gpu_checks as G
    r = G.check_post(seed=5, nc=1, th=1088, tw=1920, frac=0.05)
    assert r["candidates"] > 1500 and r["nms_exact_on_gpu_candidates"]


def test_nms_known_answers():
    import gpu_checks as G
    G.check_nms_known_answers()
