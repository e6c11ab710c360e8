"""`-m gpu`: the model-level C entry points (include/omni_amd.h: omni_detector_* / omni_captioner_*) on the MI355X — plan bundles exported
from the Python objects, loaded through the C ABI (own stream, own buffers, hipGraphs captured by the library), numpy in / numpy out,
bit-identical to the Python objects.  CPU twin on the emulation: tests/test_model_capi_emu_cpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_detector_and_captioner_bundles_match_the_python_objects(tmp_path):
    from omniparser_amd import _lib as L
    from omniparser_amd import bundle as B
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.synth import synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob, ensure_caption_checkpoint
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=0.5), device="cuda", precision="f32")
    imgs = np.stack([synthetic_screenshot(s, 1920, 1080) for s in (0, 1)])
    # the Python result of the SAME plan shape (batch 2: tile / split-K choices, hence the last bits, depend on the batch)
    from omniparser_amd.pipeline import ScreenParser
    sp = ScreenParser(det, None, processor=object(), box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640)
    want_boxes = sp.detect([torch.from_numpy(im).cuda() for im in imgs])
    want_scores = det.get_plan(1920, 1080, 640, 0.05, 0.1, 300, batch=2).out_scores.cpu().numpy()
    info = B.export_detector(det, 1920, 1080, tmp_path / "det.omniplan", imgsz=640, conf=0.05, iou=0.1, max_det=300, batch=2)
    m = L.CModel(tmp_path / "det.omniplan", "detector")
    for _ in range(3):                                   # graph replays
        boxes, scores, cls, cnt = m.infer(imgs)
    for f in range(2):
        k = int(cnt[f])
        assert k == want_boxes[f].shape[0] and k > 20
        assert np.array_equal(boxes[f, :k], want_boxes[f].numpy()) and np.array_equal(scores[f, :k], want_scores[f, :k])
    m.close()
    cap = Florence2Captioner(ensure_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    rects = [[int(v) for v in b] for b in want_boxes[0][:11].tolist()]
    rects = [[x0, y0, max(x1, x0 + 2), max(y1, y0 + 2)] for x0, y0, x1, y1 in rects]
    ids_py = cap.caption_crops(torch.from_numpy(imgs[0]).cuda(), rects, max_new_tokens=20, batch_size=8)
    B.export_captioner(cap, tmp_path / "cap.omniplan", capacity=8, max_new_tokens=20)
    c = L.CModel(tmp_path / "cap.omniplan", "captioner")
    ids = c.caption(imgs[0], rects)                      # 11 crops through an 8-row bundle: two micro-batches inside the C call
    T = ids_py.shape[1]
    assert np.array_equal(ids[:, :T], ids_py.numpy().astype(np.int32)) and (ids[:, T:] == cap.w.pad).all()
    c.close()
    print(info)
