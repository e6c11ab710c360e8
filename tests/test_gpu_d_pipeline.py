"""`-m gpu`: the two stages composed — behind the reference API (`get_som_labeled_img`), through `ScreenParser.parse_batch` with the
hand-off on the device and on the host, and the exact composition bench.py times (batch 8, full width, 768x768 crops)."""
import pytest

pytestmark = pytest.mark.gpu


def test_end_to_end_get_som_labeled_img():
    """detect -> glue -> crop -> caption behind the reference API vs the reference-equivalent CPU pipeline:
    same elements, IoU >= 0.999, captions token-exact wherever the integer crop box coincides."""
    import gpu_checks as G
    out = G.check_end_to_end(width=0.5, R=64, image_seed=1)
    assert out["n_gpu"] == out["n_ref"] and out["min_iou"] >= 0.999
    assert out["identical_crops_token_exact"] >= 0.8 * out["captioned"]


def test_parse_batch_device_handoff_equals_host_handoff(monkeypatch):
    """same frames through parse_batch with the hand-off on the device (OMNI_DEVICE_GLUE=1) and on the host (default): identical element
    lists, crop rectangles and caption ids.  (The device path launches the detector plan eagerly: a second replay of the detector
    hipGraph followed by the hand-off kernels did not complete on ROCm 7.2 — profiles/r2_notes.md — which is also why the device
    hand-off is opt-in.)"""
    import torch
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob, ensure_caption_checkpoint
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=0.5), device="cuda", precision="f32")
    cap = Florence2Captioner(ensure_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    frames = [torch.from_numpy(synthetic_screenshot(s, 1920, 1080)).cuda() for s in range(4)]
    ocr = [synthetic_ocr(s, 1920, 1080, 40) for s in range(4)]
    ocr[3] = ([], [])                                            # a frame without OCR
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("OMNI_DEVICE_GLUE", mode)
        sp = ScreenParser(det, cap, box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640)
        assert sp.device_glue == (mode == "1")
        elems, ids = sp.parse_batch(frames, ocr, return_ids=True)
        res[mode] = (elems, [[r.tolist() for r in f] for f in ids], sp.last_crops)
    assert res["1"][0] == res["0"][0] and res["1"][2] == res["0"][2] and res["1"][1] == res["0"][1]
    assert sum(len(c) for c in res["1"][2]) > 50


def test_bench_path_parity_batch8_full_width_r768():
    """The composition bench.py times (configs[2]): parse_batch over 8 frames, full-width detector, 768x768 crops packed
    across frames into 128-crop micro-batches — elements of every frame and caption ids across frame / micro-batch seams."""
    import gpu_checks as G
    out = G.check_bench_path(R=768, width=1.0, n_frames=8)
    assert out["caption_crops_checked"] >= 16 and len(out["frames_touched"]) >= 4 and len(out["micro_batches_touched"]) >= 2, out
    print(out)
