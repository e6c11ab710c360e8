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


def test_parse_stream_overlapped_equals_parse_batch():
    """ScreenParser.parse_stream with the device hand-off: batch i+1's detector graph on the detector's stream and batch i's 20 decode
    steps on the captioner's second stream (two alternating decode plans) while batch i+1 encodes — element lists, crop rectangles and
    caption ids of five batches (rotating frames, > 128 crops each, so every batch takes the merged decode) equal parse_batch's."""
    import torch
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob, ensure_caption_checkpoint
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=0.5), device="cuda", precision="f32")
    cap = Florence2Captioner(ensure_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    frames = [torch.from_numpy(synthetic_screenshot(s, 1920, 1080)).cuda() for s in range(6)]
    ocr = [synthetic_ocr(s, 1920, 1080, 40) for s in range(6)]
    sp = ScreenParser(det, cap, box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640)
    assert sp.device_glue
    batches = [([frames[(b + j) % 6] for j in range(4)], [ocr[(b + j) % 6] for j in range(4)]) for b in range(5)]
    want = []
    for f, o in batches:
        elems, ids = sp.parse_batch(f, o, return_ids=True)
        want.append((elems, [[r.tolist() for r in fr] for fr in ids], sp.last_crops))
    got = []
    for elems, ids in sp.parse_stream(iter(batches), return_ids=True):
        got.append((elems, [[r.tolist() for r in fr] for fr in ids], sp.last_crops))
    assert len(got) == 5
    for b, (w, g) in enumerate(zip(want, got)):
        assert g[2] == w[2], f"batch {b}: crop rectangles differ"
        assert g[0] == w[0], f"batch {b}: elements differ"
        assert g[1] == w[1], f"batch {b}: caption ids differ"
    assert min(sum(len(c) for c in w[2]) for w in want) > 128          # merged decode (and therefore the second stream) on every batch
    assert any(k[0] == "dec" and len(k) == 5 for k in cap._plans if isinstance(k, tuple)), "the second decode plan was never used"
