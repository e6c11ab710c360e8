"""`-m gpu`: BASELINE configs[3] — the mixed-resolution eval stream (`omniparser_amd.stream.run_stream`, what tools/stream_bench.py
times) on frames at the sizes of `stream.RESOLUTION_MIX` (1920x1080 ... 5120x2880) against the oracle pipeline, frame by frame.
Reference call being replaced: ref:eval/ss_pro_gpt4o_omniv2.py:37-51 (`get_som_labeled_img` per screenshot)."""
import pytest

pytestmark = pytest.mark.gpu


def test_stream_mixed_resolutions_elements_and_captions_vs_oracle():
    """every frame of the stream: the gathered record = the oracle's element list one for one (IoU >= 0.999, classes, order up to
    equal-score exchanges) and the oracle's greedy caption ids on every captioned icon (64x64 crops = the reference's cuda-branch crop
    size; every crop is compared, the oracle rows come from the committed cache or are computed live)."""
    import gpu_checks as G
    out = G.check_stream_parity(R=64)
    print(out)
    assert out["frames"] >= 7 and out["batches"] >= 6 and out["min_iou"] >= 0.999, out
    assert out["compared"] >= 0.97 * out["captioned"] and out["captioned"] >= 100, out      # not compared = integer crop rectangle one pixel apart


def test_stream_768_crops_first_frames_vs_oracle():
    """the same stream at the parity crop size (768x768 = the reference's CPU path) on its first two frames: elements one for one and
    caption ids token-exact (CPU Florence-2 at 768x768 costs seconds per crop: two frames, cached rows)."""
    import gpu_checks as G
    out = G.check_stream_parity(R=768, n_frames=2)
    print(out)
    assert out["frames"] == 2 and out["min_iou"] >= 0.999 and out["compared"] >= 0.97 * out["captioned"] and out["captioned"] >= 20, out
