"""`-m gpu`: the two stages composed — behind the reference API (`get_som_labeled_img`) and in the exact composition bench.py times
(`ScreenParser.parse_batch`, batch 8, full width, 768x768 crops)."""
import pytest

pytestmark = pytest.mark.gpu


def test_end_to_end_get_som_labeled_img():
    """detect -> glue -> crop -> caption behind the reference API vs the reference-equivalent CPU pipeline:
    same elements, IoU >= 0.999, captions token-exact wherever the integer crop box coincides."""
    import gpu_checks as G
    out = G.check_end_to_end(width=0.5, R=64, image_seed=1)
    assert out["n_gpu"] == out["n_ref"] and out["min_iou"] >= 0.999
    assert out["identical_crops_token_exact"] >= 0.8 * out["captioned"]


def test_bench_path_parity_batch8_full_width_r768():
    """The composition bench.py times (configs[2]): parse_batch over 8 frames, full-width detector, 768x768 crops packed
    across frames into 128-crop micro-batches — elements of every frame and caption ids across frame / micro-batch seams."""
    import gpu_checks as G
    out = G.check_bench_path(R=768, width=1.0, n_frames=8)
    assert out["caption_crops_checked"] >= 16 and len(out["frames_touched"]) >= 4 and len(out["micro_batches_touched"]) >= 2, out
    print(out)
