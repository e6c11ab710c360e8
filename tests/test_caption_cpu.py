"""CPU check of the Florence-2 plan lowering: the op lists built by omniparser_amd/florence.py, run by
the torch interpreter, must reproduce transformers' native Florence-2 (image features, encoder output,
greedy ids incl. NoRepeatNGram / ForcedBOS / ForcedEOS) on the reference's cuda-branch shape (64x64)."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def ckpt():
    from tools.make_weights import ensure_caption_checkpoint, build_random_captioner
    d = ensure_caption_checkpoint(0)
    return d, build_random_captioner(0)


def test_florence_plan_matches_transformers_r64(ckpt):
    import caption_checks as CC
    d, model = ckpt
    g = torch.Generator().manual_seed(5)
    pix = torch.randn(2, 3, 64, 64, generator=g)
    feats, enc, ids = CC.hf_reference(model, pix, 20)
    cap, cp = CC.build_cpu_plans(d, 2, 64)
    f2, e2, i2 = CC.run_interp(cap, cp, pix, 20)
    assert (f2 - feats).abs().max() < 2e-3 * feats.abs().max()
    assert (e2 - enc).abs().max() < 2e-3 * enc.abs().max()
    assert torch.equal(i2[:, : ids.shape[1]], ids), (i2, ids)


def test_crop_preprocess_oracle_invariants():
    from oracle import preprocess_ref as PR
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    assert np.array_equal(PR.cv2_resize_linear(img, (64, 64)), img)            # identity at 64 -> 64
    const = np.full((37, 91, 3), 123, dtype=np.uint8)
    assert (PR.cv2_resize_linear(const, (64, 64)) == 123).all()                # constants are preserved
    big = rng.integers(0, 256, size=(128, 128, 3), dtype=np.uint8).astype(np.int64)
    area = (big[0::2, 0::2] + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2
    assert np.array_equal(PR.cv2_resize_linear(big.astype(np.uint8), (64, 64)), area.astype(np.uint8))  # exact 2x == INTER_AREA
    up = PR.cv2_resize_linear(img[:8, :8], (64, 64))
    assert np.array_equal(up[0, 0], img[0, 0]) and np.array_equal(up[-1, -1], img[7, 7])   # corners clamp
    pv = PR.caption_pixel_values(img, (0, 0, 64, 64), 64, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    assert pv.shape == (64, 64, 3) and pv.dtype == np.float32
