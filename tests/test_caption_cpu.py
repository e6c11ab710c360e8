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


def test_activation_reuse_plan_matches_transformers_and_halves_the_scratch_bytes(ckpt, monkeypatch, tmp_path):
    """Florence2Captioner.reuse_activations (default since round 4): the scratch tensors of a DaViT stage are released at its end
    and back the tensors allocated later (PlanBuilder.release / free list).  The interpreter runs the op list on those aliased
    buffers: a live-range mistake shows as wrong features / ids against transformers.  Bytes: < 0.5 x the default plan's scratch."""
    import caption_checks as CC
    from omniparser_amd import florence as FL, planner

    def scratch_bytes(cp):
        seen = {}
        for t in cp.pb.keep:
            if isinstance(t, torch.Tensor) and planner._TENSORS.get(t.data_ptr(), (None, "const"))[1] != "const":
                seen[t.data_ptr()] = t.numel() * t.element_size()
        return sum(seen.values())

    d, model = ckpt
    g = torch.Generator().manual_seed(5)
    pix = torch.randn(2, 3, 64, 64, generator=g)
    feats, enc, ids = CC.hf_reference(model, pix, 20)
    assert FL.Florence2Captioner.reuse_activations is True
    monkeypatch.setattr(FL.Florence2Captioner, "reuse_activations", False)
    cap0, cp0 = CC.build_cpu_plans(d, 2, 64)                 # every stage owns its buffers
    monkeypatch.undo()
    cap, cp = CC.build_cpu_plans(d, 2, 64)                   # the default
    assert cp.pb.reuse and cp.pb.reused_bytes > 0 and not cp0.pb.reuse
    assert [(o.kind, tuple(o.i)) for o in cp.encode_plan.ops] == [(o.kind, tuple(o.i)) for o in cp0.encode_plan.ops]   # same ops, other addresses
    assert scratch_bytes(cp) < 0.5 * scratch_bytes(cp0), (scratch_bytes(cp), scratch_bytes(cp0))
    # the tensors that outlive the encode plan (its input, what the decode side and the callers read) were carved after the releases:
    # none of them may share a byte with another one
    live = [cp.x_in, cp.img_feat, cp.enc_out, cp.vision_out] + list(cp.cross_kv)
    spans = sorted((v.t.data_ptr(), v.t.data_ptr() + v.t.numel() * v.t.element_size()) for v in live)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), spans
    f2, e2, i2 = CC.run_interp(cap, cp, pix, 20)
    assert (f2 - feats).abs().max() < 2e-3 * feats.abs().max()
    assert (e2 - enc).abs().max() < 2e-3 * enc.abs().max()
    assert torch.equal(i2[:, : ids.shape[1]], ids), (i2, ids)
    f0, e0, i0 = CC.run_interp(cap0, cp0, pix, 20)          # same ops on the same data: bit for bit what the plan without reuse computes
    assert torch.equal(f2, f0) and torch.equal(e2, e0) and torch.equal(i2, i0)
    # plan export (model-level C entry points): pointers into carved tensors resolve against the registered blocks they were carved from
    from omniparser_amd import bundle as BN
    whole = lambda v: (v.t, 0, v.t.numel() * v.t.element_size())
    info = [BN.write_bundle(tmp_path / f"reuse{k}.omniplan", {"encode": c.encode_plan.ops}, {"x_in": whole(c.x_in), "enc_out": whole(c.enc_out)}, {})
            for k, c in enumerate((cp0, cp))]
    assert info[1]["ops"] == info[0]["ops"] and info[1]["device_bytes"] < info[0]["device_bytes"]
    rb = BN.read_bundle(tmp_path / "reuse1.omniplan")
    tens, (k, off, nb) = rb["tensors"], rb["named"]["enc_out"]
    assert off + nb <= tens[k][0] and any(o != 0 for _, _, ptrs, _, _ in rb["plans"]["encode"] for t, o in ptrs if t >= 0)


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
