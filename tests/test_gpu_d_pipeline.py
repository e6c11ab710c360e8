"""`-m gpu`: the two stages composed behind the reference API (`get_som_labeled_img`).  The exact composition bench.py times
(`ScreenParser.parse_batch`, batch 8, full width, 768x768 crops) is tests/test_gpu_z_bench_path.py — the longest test, run last."""
import pytest

pytestmark = pytest.mark.gpu


def test_end_to_end_get_som_labeled_img():
    """detect -> glue -> crop -> caption behind the reference API vs the reference-equivalent CPU pipeline:
    same elements, IoU >= 0.999, captions token-exact wherever the integer crop box coincides."""
    import gpu_checks as G
    out = G.check_end_to_end(width=0.5, R=64, image_seed=1)
    assert out["n_gpu"] == out["n_ref"] and out["min_iou"] >= 0.999
    assert out["identical_crops_token_exact"] >= 0.95 * out["captioned"], out      # the rest: integer crop rectangles one pixel apart (counted)


def test_tiled_4k_end_to_end_captions_token_exact():
    """BASELINE configs[4]: 3840x2160 -> tiled detection -> ~200 crops in 64-crop micro-batches: elements / crop rectangles = the
    reference hand-off of the tiled boxes, caption ids of every crop = the CPU oracle's (greedy, token-exact)."""
    import gpu_checks as G
    out = G.check_tiled_captions(width=0.5, R=64)
    assert out["crops"] >= 130 and out["micro_batches"] >= 3 and out["compared"] == out["crops"], out       # every crop, whatever its margin
    print(out)


def test_tiled_4k_end_to_end_captions_token_exact_r768():
    """BASELINE configs[4] at the crop size it names: the same 3840x2160 frame, tiled detection, 64-crop caption micro-batches at
    768x768 (the reference's CPU-path crop size) — caption ids of every crop = the CPU oracle's (oracle rows from the committed cache,
    generated in the CPU container: tests/golden/gen_oracle_cache.py `tiled`)."""
    import gpu_checks as G
    out = G.check_tiled_captions(width=0.5, R=768)
    # EVERY crop is compared (round 5 skipped the 8 of 137 whose oracle margin is below 1e-3); a mismatch is a failure unless the crop's
    # margin is below 2e-4 AND the f64 oracle does not side with the f32 oracle (gpu_checks.CaptionTally)
    assert out["crops"] >= 130 and out["micro_batches"] >= 3 and out["compared"] == out["crops"], out
    print(out)
