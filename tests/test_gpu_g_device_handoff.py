"""`-m gpu`: the device-side detect -> caption hand-off (csrc/glue_ops.hip; the default of `ScreenParser`, OMNI_DEVICE_GLUE=0 = host
twin) — the kernel against the fixtures recorded from the reference's own functions and against the host twin, then through
`ScreenParser.parse_batch`, then 30 consecutive replays of the detector + hand-off graph."""
import pytest

pytestmark = pytest.mark.gpu


def test_device_handoff_matches_reference_fixtures_and_host_twin():
    """OMNI_OP_GLUE: overlap removal + ordering + crop rectangles on the device, exact vs the reference fixtures and vs the host path."""
    import gpu_checks as G
    r = G.check_glue()
    assert r["fixture_cases"] >= 12 and r["random_trials"] >= 40


def test_parse_batch_device_handoff_equals_host_handoff(monkeypatch):
    """same frames through parse_batch with the hand-off on the device (default: detector + hand-off ops captured as ONE hipGraph) and
    on the host (OMNI_DEVICE_GLUE=0): identical element lists, crop rectangles and caption ids."""
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


def test_handoff_graph_replays_stay_correct():
    """30 consecutive replays of the detector + hand-off graph over rotating frames: every replay's crop rectangles equal the host
    twin's for the same detector boxes (round 2 saw the SECOND replay stall while the detector graph still held a memset node)."""
    import torch
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=0.5), device="cuda", precision="f32")
    sp = ScreenParser(det, None, processor=object(), box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640)
    assert sp.device_glue and det.use_graph
    frames = [torch.from_numpy(synthetic_screenshot(s, 1920, 1080)).cuda() for s in range(4)]
    ocr = [synthetic_ocr(s, 1920, 1080, 40) for s in range(4)]
    for it in range(30):
        rot = it % 4
        fr, oc = frames[rot:] + frames[:rot], ocr[rot:] + ocr[:rot]
        dp, gs, ocr_els, counts = sp.detect_glue(fr, oc)
        assert gs.plan.captured
        boxes, kc = dp.out_boxes.cpu(), dp.out_count.cpu()
        for f in range(4):
            el, cr = sp.glue(boxes[f, : int(kc[f])], 1920, 1080, oc[f][1], oc[f][0])
            assert [list(c) for c in cr] == gs.crops[f, : int(counts[f, 1])].tolist(), (it, f)
