"""The model-level C entry points (include/omni_amd.h: omni_detector_* / omni_captioner_*, csrc/model_api.hip) on the host emulation:
a detector and a captioner are exported as plan bundles (omniparser_amd/bundle.py), loaded through the C ABI with numpy buffers only,
and must return what the Python objects return — bit for bit, it is the same op list on the same kernels.  The MI355X twin is
tests/test_gpu_i_model_capi.py."""
import numpy as np
import torch

from omniparser_amd import _lib as L


def test_bundle_file_roundtrip(emu, tmp_path):
    """what export writes is what the reader reads: every op, every pointer as (tensor, offset), named tensors, scalars."""
    from omniparser_amd import bundle as B
    from omniparser_amd.planner import PlanBuilder, View
    pb = PlanBuilder("cpu", L.F32)
    x = pb.alloc(1, 4, 4, 32, zero=True)
    y = pb.alloc(1, 4, 4, 64)
    w = pb.pack_weight(torch.randn(64, 32, 1, 1))
    pb.conv(x, w, torch.randn(64), y, 1, act=L.ACT_SILU)
    info = B.write_bundle(tmp_path / "t.omniplan", {"p": pb.ops}, {"x": (x.t, 0, x.t.numel() * 4), "y": (y.t, 64, 128)}, {"answer": 42, "neg": -7})
    d = B.read_bundle(tmp_path / "t.omniplan")
    assert d["ints"] == {"answer": 42, "neg": -7} and set(d["named"]) == {"x", "y"} and d["named"]["y"][1:] == (64, 128)
    (kind, dtype, ptrs, ii, ff), = d["plans"]["p"]
    assert kind == L.OP_CONV and ii[3] == 32 and ii[12] == 64 and ii[15] == L.ACT_SILU
    assert [p[0] >= 0 for p in ptrs] == [bool(v) for v in pb.ops[0].p]
    roles = sorted(t[1] for t in d["tensors"])
    assert roles.count(2) >= 2 and roles.count(1) == 1          # weight + bias constants, one zero-initialised input
    assert info["ops"] == {"p": 1}


def test_detector_bundle_through_c_entry_points(emu, tmp_path, monkeypatch):
    from omniparser_amd import bundle as B
    from omniparser_amd.synth import synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob
    monkeypatch.setenv("OMNI_VERIFY_IMPORT", "0")
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=0.25), device="cuda", precision="f32")
    img = synthetic_screenshot(0, 640, 480)
    res = det.predict(img, conf=0.05, imgsz=320, iou=0.1, max_det=300)[0]
    info = B.export_detector(det, 640, 480, tmp_path / "det.omniplan", imgsz=320, conf=0.05, iou=0.1, max_det=300, batch=1)
    m = L.CModel(tmp_path / "det.omniplan", "detector")
    assert (m.int("batch"), m.int("img_w"), m.int("img_h"), m.int("max_det")) == (1, 640, 480, 300)
    boxes, scores, cls, cnt = m.infer(img[None])
    k = int(cnt[0])
    assert k == res.boxes.xyxy.shape[0] and k > 5
    assert np.array_equal(boxes[0, :k], res.boxes.xyxy.numpy()) and np.array_equal(scores[0, :k], res.boxes.conf.numpy())
    assert info["ops"]["detect"] > 250
    m.close()


def test_captioner_bundle_through_c_entry_points(emu, tmp_path):
    from omniparser_amd import bundle as B
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.synth import synthetic_screenshot
    from tools.make_weights import ensure_caption_checkpoint
    from conftest import small_vocab_caption_checkpoint        # 8192-row token table: see its docstring
    cap = Florence2Captioner(small_vocab_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    img = synthetic_screenshot(1, 640, 480)
    boxes = [[10, 20, 60, 70], [300, 200, 340, 260], [500, 100, 620, 140]]
    max_new = 2                                   # the emulation pays ~15 G multiply-adds per lm_head step
    want = cap.caption_crops(torch.from_numpy(img), boxes, max_new_tokens=max_new, batch_size=2)
    B.export_captioner(cap, tmp_path / "cap.omniplan", capacity=2, max_new_tokens=max_new)
    m = L.CModel(tmp_path / "cap.omniplan", "captioner")
    assert (m.int("capacity"), m.int("R"), m.int("T")) == (2, 64, max_new + 1)
    ids = m.caption(img, boxes)                   # 3 crops through a 2-row plan: two micro-batches inside the C call
    T = want.shape[1]
    assert np.array_equal(ids[:, :T], want.numpy().astype(np.int32))
    assert (ids[:, T:] == cap.w.pad).all()
    m.close()
