"""`-m gpu`, LAST file of the suite (minutes of CPU oracle time; `pytest -x` must not let it hide the other files): the exact
composition bench.py times — `ScreenParser.parse_batch`, batch 8, full-width detector, 768x768 crops in packed micro-batches."""
import pytest

pytestmark = pytest.mark.gpu


def test_bench_path_parity_batch8_full_width_r768():
    """The composition bench.py times (configs[2]): parse_batch over 8 frames, full-width detector, 768x768 crops packed
    across frames into 128-crop micro-batches — elements of every frame and caption ids across frame / micro-batch seams."""
    import gpu_checks as G
    out = G.check_bench_path(R=768, width=1.0, n_frames=8, min_exact=8)
    assert out["caption_crops_checked"] >= 16 and len(out["frames_touched"]) >= 4 and len(out["micro_batches_touched"]) >= 2, out
    # the benched frames (synth.BENCH_SEEDS) were chosen tie-free on the CPU oracle: EVERY frame is compared element for element with
    # the oracle's own list, and the pipelined composition reproduced parse_batch on all three batches
    assert out["exact_frames"] == 8 and out["stream_batches_equal_parse_batch"] == 3, out
    print(out)
