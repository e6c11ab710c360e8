"""`-m gpu`: Florence-2 captioner kernels and the whole caption / end-to-end path vs CPU references."""
import pytest

from omniparser_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [L.F32, L.F16])
def test_caption_kernels_vs_interpreter(dtype):
    import gpu_checks as G
    G.check_caption_ops(dtype)


def test_captioner_token_exact_r64():
    """reference cuda-branch shape (64x64 crops, 5 image tokens): greedy ids == transformers CPU."""
    import gpu_checks as G
    out, _ = G.check_captioner(R=64, n=5)
    assert out["ids_equal"], out
    assert out["feat_rel_err"] < 1e-4 and out["enc_rel_err"] < 1e-4


def test_captioner_token_exact_r768():
    """reference CPU-branch shape (768x768 crops, 577 image tokens) — the parity target."""
    import gpu_checks as G
    out, _ = G.check_captioner(R=768, n=2)
    assert out["ids_equal"], out
    assert out["feat_rel_err"] < 3e-4 and out["enc_rel_err"] < 3e-4


def test_end_to_end_get_som_labeled_img():
    """detect -> glue -> crop -> caption behind the reference API vs the reference-equivalent CPU pipeline:
    same elements, IoU >= 0.999, captions token-exact wherever the integer crop box coincides."""
    import gpu_checks as G
    out = G.check_end_to_end(width=0.5, R=64, image_seed=1)
    assert out["n_gpu"] == out["n_ref"] and out["min_iou"] >= 0.999
    assert out["identical_crops_token_exact"] >= 0.8 * out["captioned"]
