"""`-m gpu` detector tests rehearsed on the CPU: the same check_detector / assert_detector_frame code, with the oracle's float64
evaluation standing in for the device (tests/rehearsal_device.py).  Catches rules that rounding noise alone would trip — the GPU suite
cannot be iterated on cheaply — and pins tools/make_weights.py::EXACT_FRAMES: every listed frame must pass with exact=True."""
import pytest


@pytest.fixture
def rehearsal(monkeypatch):
    import omniparser_amd.util.yolov9 as Y
    from rehearsal_device import RehearsalDetector

    def use(width):
        monkeypatch.setattr(Y, "YOLOv9Detector", lambda model_path, device="cuda", precision="f32": RehearsalDetector(model_path, width))
    return use


def test_half_width_exact_frames_and_a_tie_frame(rehearsal):
    import gpu_checks as G
    from tools.make_weights import EXACT_FRAMES
    rehearsal(0.5)
    out, _ = G.check_detector(width=0.5, image_seeds=EXACT_FRAMES[(0.5, 640)], imgsz=640)
    for rec in out["images"]:
        G.assert_detector_frame(rec, exact=True)
        assert rec["zero_area_boxes"] > 0            # the case that made IoU-only matching report dozens of "lost" boxes
    out, _ = G.check_detector(width=0.5, image_seeds=(0,), imgsz=640)
    assert out["images"][0]["score_ties"] + out["images"][0]["near_ties"] >= 1
    G.assert_detector_frame(out["images"][0])


def test_quarter_width_smoke_frame_and_native_resolution(rehearsal):
    import gpu_checks as G
    from tools.make_weights import EXACT_FRAMES
    rehearsal(0.25)
    out, _ = G.check_detector(width=0.25, image_seeds=EXACT_FRAMES[(0.25, 320)][:1], imgsz=320, iw=640, ih=480, with_f64=True)
    G.assert_detector_frame(out["images"][0], exact=True)
    out, _ = G.check_detector(width=0.25, image_seeds=EXACT_FRAMES[(0.25, 640)], imgsz=640)
    G.assert_detector_frame(out["images"][0], exact=True)
    # 1088x1920 (scale_img=True): since the v5 stand-in the fixed head epsilon holds here too (its own f32-vs-f64 difference is
    # <= 3e-5), the scanned frame must pass with exact=True: the oracle's 300 boxes one for one
    out, _ = G.check_detector(width=0.25, image_seeds=EXACT_FRAMES[(0.25, "native")][:1], imgsz=(1080, 1920), with_f64=True)
    rec = out["images"][0]
    assert rec["n_ref"] == 300 and max(n[0] for n in rec["oracle_noise(cls,dist,gpu_vs_f64)"]) <= 3e-5, rec
    G.assert_detector_frame(rec, exact=True)


def test_end_to_end_check_rehearsed(rehearsal, monkeypatch):
    """tests/gpu_checks.py::check_end_to_end (get_som_labeled_img on the device path vs the oracle pipeline) with the float64 detector
    and the transformers captioner standing in for the device: its pairing / caption comparison must accept rounding noise."""
    import gpu_checks as G
    import omniparser_amd.florence as F
    from tools.make_weights import build_random_captioner
    rehearsal(0.5)
    model = build_random_captioner(0)
    monkeypatch.setattr(F, "Florence2Captioner", lambda cdir, device, precision="f32", resolution=64: G._OracleCaptioner(model, resolution))
    out = G.check_end_to_end(width=0.5, R=64, image_seed=1)
    assert out["n_gpu"] == out["n_ref"] and out["min_iou"] >= 0.999
    assert out["identical_crops_token_exact"] >= 0.8 * out["captioned"] and out["captioned"] > 20


def test_bench_path_check_rehearsed(rehearsal, monkeypatch):
    """tests/gpu_checks.py::check_bench_path at a size the CPU finishes (2 frames, half width, 64x64 crops, 32-crop micro-batches):
    frame-boundary / micro-batch-boundary crop selection, margin bookkeeping, crop-tensor indexing and the element rules all run."""
    import gpu_checks as G
    import omniparser_amd.florence as F
    import omniparser_amd.pipeline as P
    import tools.make_weights as MW
    from rehearsal_device import RehearsalCaptioner, RehearsalParser
    rehearsal(0.5)
    model = MW.build_random_captioner(0)
    monkeypatch.setattr(F, "Florence2Captioner",
                        lambda cdir, device, precision="f32", resolution=64: RehearsalCaptioner(G._OracleCaptioner(model, resolution), resolution))
    monkeypatch.setattr(P, "ScreenParser", RehearsalParser)
    monkeypatch.setattr(G, "DEV", "cpu")
    out = G.check_bench_path(R=64, width=0.5, n_frames=2, caption_pairs=((0, 1),), per_side=3, boundary=2, min_exact=1)
    assert out["exact_frames"] == 1 and out["score_ties"][0] >= 1 and out["crop_coords_off_by_one"] == 0
    assert out["caption_crops_checked"] >= 8 and out["frames_touched"] == [0, 1] and len(out["micro_batches_touched"]) >= 2
    assert all(out["crop_tensor_bitwise"]) and out["matched_fraction"] == [1.0, 1.0]
    # the non-curated frame set of tests/test_gpu_z_bench_path.py (explicit seeds, detector + hand-off statements only)
    out = G.check_bench_path(R=64, width=0.5, n_frames=2, seeds=(5, 6), min_exact=0, detector_only=True)
    assert out["seeds"] == [5, 6] and all(out["expected_from_device_candidates"]) and "caption_crops_checked" not in out
    assert min(out["matched_fraction"]) >= 0.95


def test_rehearse_reference_image_end_to_end(rehearsal, monkeypatch):
    """tests/test_gpu_j_reference_images.py's check on word.png (1919x1079 RGBA, the reference's own fixture): RGBA -> RGB, odd-size
    letterbox, real-image crops, the capped CPU captioner (captions beyond the budget are not compared) — at 64x64 crops and quarter
    width, the oracle standing in for both device models."""
    from pathlib import Path
    from PIL import Image
    import gpu_checks as G
    from omniparser_amd import florence as F
    from omniparser_amd.synth import synthetic_ocr
    from tools.make_weights import build_random_captioner
    rehearsal(0.25)
    model = build_random_captioner(0)
    monkeypatch.setattr(F, "Florence2Captioner", lambda cdir, device, precision="f32", resolution=64: G._OracleCaptioner(model, resolution))
    img = Image.open(Path(__file__).parent / "golden" / "ref_imgs" / "word.png")
    out = G.check_end_to_end(width=0.25, R=64, image=img, ocr=synthetic_ocr(7, img.size[0], img.size[1], 40), max_crops_checked=3)
    assert out["size"] == [1919, 1079] and out["n_gpu"] == out["n_ref"] and out["min_iou"] >= 0.999
    assert out["captioned"] == 3 and out["identical_crops_token_exact"] == 3, out
