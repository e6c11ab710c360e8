"""The product's DEVICE PATH — plan builders, blob importer, detector / captioner objects, ScreenParser — executed on the host
emulation of its own kernels (tests/emu) and held to the `-m gpu` parity checks at sizes a CPU finishes in about a minute:
quarter-width YOLOv9-E stand-in at a 320x320 network input, Florence-2-base-shaped captioner at 64x64 crops.  The checks are the
ones the MI355X runs (tests/gpu_checks.py); what the hardware alone decides (timing, hipGraph replay, LDS capacity) stays with them."""
import pytest
import torch

from omniparser_amd import _lib as L


def test_detector_through_reference_api_matches_oracle_box_for_box(emu):
    """YOLOv9Detector.predict (import-verified blob -> plan -> letterbox / network / decode / NMS kernels) vs oracle.detector_ref on a
    frame of tools/make_weights.py::EXACT_FRAMES: byte-exact input, heads within the fixed epsilon, identical candidates, NMS
    bit-exact, final boxes one for one."""
    import gpu_checks as G
    from tools.make_weights import EXACT_FRAMES
    out, det = G.check_detector(width=0.25, image_seeds=EXACT_FRAMES[(0.25, 320)][:1], imgsz=320, iw=640, ih=480)
    assert det.import_rel_err is not None and det.import_rel_err < 1e-3           # load-time proof of the imported blob ran on the kernels
    G.assert_detector_frame(out["images"][0], exact=True)
    assert out["ops"] > 250


def test_captioner_token_exact_r64(emu, monkeypatch):
    """Florence2Captioner.generate (DaViT tower, projector, BART encoder / decoder with KV cache, greedy loop: every captioner kernel)
    vs transformers on the CPU: image features, encoder output, greedy ids.  Three decode steps: a single-row lm_head (768 x 51289,
    padded to a 128-row tile) costs the emulation 15 G multiply-adds per step; the 21-token loops run on the MI355X
    (tests/test_gpu_b_caption_model.py) and in the plan interpreter (tests/test_caption_cpu.py)."""
    import caption_checks as CC
    import gpu_checks as G
    from omniparser_amd.florence import Florence2Captioner
    from tools.make_weights import build_random_captioner, ensure_caption_checkpoint
    max_new = 3
    pix = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(75))
    feats, enc, ids = CC.hf_reference(build_random_captioner(0), pix, max_new)
    cap = Florence2Captioner(ensure_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    got = cap.generate(pixel_values=pix, max_new_tokens=max_new)
    cp = cap.plans(cap.bucket(1), 64, max_new)
    assert G.rel_err(cp.img_feat.t[:1, :, 0, :].float(), feats) < 1e-4
    assert G.rel_err(cp.enc_out.t[:1, :, 0, :].float(), enc) < 1e-4
    assert got.shape == ids.shape and torch.equal(got, ids), (got, ids)
    # default plan: attention / channel attention write format B themselves and x + dwconv(x) -> LayerNorm is ONE kernel in stages
    # 0-2; with both switched off (the round-2 composition: separate dwconv / LayerNorm kernels, f32 attention output converted in
    # place) the same features come out — encode pass only
    assert any(op.kind == L.OP_DWCONV3_LN for op in cp.encode_plan.ops)
    monkeypatch.setattr(Florence2Captioner, "attn_split_out", False)
    monkeypatch.setattr(Florence2Captioner, "fuse_dwln", False)
    cap2 = Florence2Captioner(ensure_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    cap2._wcache = cap._wcache                  # same checkpoint: packed weights are shared, not packed again
    cp2 = cap2.plans(1, 64, max_new)
    assert not any(op.kind == L.OP_DWCONV3_LN for op in cp2.encode_plan.ops)
    assert sum(op.kind == L.OP_SPLIT_CONVERT for op in cp2.encode_plan.ops) > sum(op.kind == L.OP_SPLIT_CONVERT for op in cp.encode_plan.ops)
    with torch.inference_mode():
        cp2.reset()
        cp2.x_in.t[:1, :, :, :3] = pix.permute(0, 2, 3, 1)
        cp2.encode_plan.run(cap2.stream)
    assert G.rel_err(cp2.img_feat.t[:1, :, 0, :].float(), feats) < 1e-4
    assert G.rel_err(cp2.enc_out.t[:1, :, 0, :].float(), enc) < 1e-4
    # the default plan's attention kernels at 64x64: every stage has cut windows (16, 8, 4, 2 tokens per side) and token counts
    # 256, 64, 16, 4 per channel-attention group (covered by the feature comparison above).
    # Florence2Captioner.reuse_activations (default): the same kernels on aliased scratch buffers (a stage's tensors back the later
    # stages') — the same ops in the same order on the same data: features and encoder output bit for bit those of a plan whose
    # stages each own their buffers
    monkeypatch.undo()
    cp1 = cap.plans(1, 64, max_new)              # the default composition at the same row count (tile choices follow the row count)
    monkeypatch.setattr(Florence2Captioner, "reuse_activations", False)
    cap4 = Florence2Captioner(ensure_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    cap4._wcache = cap._wcache                  # same checkpoint: packed weights are shared, not packed again
    cp4 = cap4.plans(1, 64, max_new)
    assert cp1.pb.reused_bytes > 0 and cp4.pb.reused_bytes == 0
    with torch.inference_mode():
        for c, p in ((cap, cp1), (cap4, cp4)):
            p.reset()
            p.x_in.t[:1, :, :, :3] = pix.permute(0, 2, 3, 1)
            p.encode_plan.run(c.stream)
    assert torch.equal(cp4.img_feat.t, cp1.img_feat.t) and torch.equal(cp4.enc_out.t, cp1.enc_out.t)
    assert all(torch.equal(a.t, b.t) for a, b in zip(cp4.cross_kv, cp1.cross_kv))


def test_parse_batch_device_handoff_equals_host_handoff(emu, monkeypatch):
    """ScreenParser.parse_batch with the detect -> caption hand-off on the device (the default: detector + hand-off ops in one plan, glue kernel,
    crop rectangles that never visit the host, packed caption micro-batches) against the default host hand-off of the same frames:
    identical element tables and crop rectangles; captions present on every icon without OCR text."""
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob, ensure_caption_checkpoint
    monkeypatch.setenv("OMNI_VERIFY_IMPORT", "0")                 # proven in the detector test above
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=0.25), device="cuda", precision="f32")
    from conftest import small_vocab_caption_checkpoint        # 8192-row token table: see its docstring
    cap = Florence2Captioner(small_vocab_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    frames = [torch.from_numpy(synthetic_screenshot(s, 640, 480)) for s in (0, 2)]
    ocr = [synthetic_ocr(0, 640, 480, 12), ([], [])]                                  # second frame without OCR
    kw = dict(box_threshold=0.85, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=320, batch_size=8)     # (v5 stand-in: 0.85 leaves a handful of crops)
    monkeypatch.setenv("OMNI_DEVICE_GLUE", "1")
    sp = ScreenParser(det, cap, **kw)
    sp.max_new_tokens = 1                                          # the decode loop is covered by the captioner test
    assert sp.device_glue
    elems, ids = sp.parse_batch(frames, ocr, return_ids=True)
    crops = sp.last_crops
    monkeypatch.setenv("OMNI_DEVICE_GLUE", "0")
    sp0 = ScreenParser(det, cap, **kw)
    assert not sp0.device_glue
    boxes = sp0.detect(frames)
    n_crops = 0
    for f in range(2):
        el, cr = sp0.glue(boxes[f], 640, 480, ocr[f][1], ocr[f][0])
        assert cr == crops[f]
        assert len(el) == len(elems[f])
        k = 0
        for a, b in zip(elems[f], el):
            assert (a["type"], a["bbox"], a["source"], a["interactivity"]) == (b["type"], b["bbox"], b["source"], b["interactivity"])
            if b["content"] is None:
                assert isinstance(a["content"], str) and len(ids[f][k]) >= 1     # captioned from the device-side rectangles
                k += 1
            else:
                assert a["content"] == b["content"]
        assert k == len(cr)
        n_crops += k
    assert 4 <= n_crops <= 40    # one packed micro-batch (or a few) spanning both frames (the seams between micro-batches: tests/test_gpu_d_pipeline.py)


def test_parse_stream_pipeline_equals_parse_batch(emu, monkeypatch):
    """ScreenParser.parse_stream (detector + hand-off of batch i+1 on a helper thread while batch i is captioned; caption work of
    batch i+1 queued before the read-back of batch i) returns what parse_batch returns, batch by batch, in order.  Real detector,
    hand-off and crop kernels on the emulation (two threads launching concurrently); the encode / decode plans are replaced by a
    fingerprint of each crop's pixel tensor, which pins crop -> caption slot -> element across the pipeline's hand-overs."""
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob, ensure_caption_checkpoint
    monkeypatch.setenv("OMNI_VERIFY_IMPORT", "0")
    monkeypatch.setenv("OMNI_DEVICE_GLUE", "0")
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=0.25), device="cuda", precision="f32")
    from conftest import small_vocab_caption_checkpoint        # 8192-row token table: see its docstring
    cap = Florence2Captioner(small_vocab_caption_checkpoint(0), "cuda", precision="f32", resolution=64)

    def fingerprint(cp, n, max_new, defer=False):
        x = cp.x_in.t[:n, :, :, :3].double()
        key = (x.sum((1, 2, 3)) * 977.0 + x[:, ::7, ::5].sum((1, 2, 3)) * 131.0).abs()
        return torch.stack([torch.full((n,), 0, dtype=torch.long), 100 + (key.long() % 40000), torch.full((n,), 2, dtype=torch.long)], 1).int()
    monkeypatch.setattr(cap, "_run", fingerprint)
    # one micro-batch per batch (batch_size 32 >= the crops of two frames): more would take the MERGED decode — real decoder kernels on
    # the emulation, minutes — which test_merged_decode_equals_per_micro_batch_decode covers on its own
    sp = ScreenParser(det, cap, box_threshold=0.5, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=320, batch_size=32)
    batches = [([torch.from_numpy(synthetic_screenshot(s, 640, 480)) for s in seeds], [synthetic_ocr(s, 640, 480, 10) for s in seeds])
               for seeds in ((0, 1), (2,), (3, 0))]
    want, crops_want = [], []
    for f, o in batches:
        want.append(sp.parse_batch(f, o, return_ids=True, pad_to=2))
        crops_want.append(sp.last_crops)
    got = []
    crops_got = []
    for res in sp.parse_stream(iter(batches), return_ids=True, pad_to=2):
        got.append(res)
        crops_got.append(sp.last_crops)
    assert len(got) == 3 and crops_got == crops_want
    for (el_w, ids_w), (el_g, ids_g) in zip(want, got):
        assert el_g == el_w
        assert [[r.tolist() for r in f] for f in ids_g] == [[r.tolist() for r in f] for f in ids_w]
    assert sum(len(c) for b in crops_want for c in b) >= 12
    assert len({tuple(r.tolist()) for _, ids in want for f in ids for r in f}) >= 10        # the fingerprints tell crops apart


def test_parse_stream_device_handoff_equals_parse_batch(emu, monkeypatch):
    """parse_stream with the device hand-off (the default): detector tables snapshotted per batch, merged decode on the second
    stream over two alternating decode plans — batch by batch what parse_batch returns.  Real detector, hand-off and crop kernels
    on the emulation; encode / decode replaced by a fingerprint of each crop's pixel tensor carried through the decode plan's rows
    (so a wrong slot, a stale snapshot or an overwritten crop table would show)."""
    from types import SimpleNamespace
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob, ensure_caption_checkpoint
    monkeypatch.setenv("OMNI_VERIFY_IMPORT", "0")
    monkeypatch.setenv("OMNI_DEVICE_GLUE", "1")
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=0.25), device="cuda", precision="f32")
    from conftest import small_vocab_caption_checkpoint        # 8192-row token table: see its docstring
    cap = Florence2Captioner(small_vocab_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    slots_used = []

    def fake_plans(B, R, max_new, slot=0):
        key = ("fake", slot)
        if key not in cap._plans:
            cap._plans[key] = SimpleNamespace(fp=torch.zeros(256, 3, dtype=torch.int32), free_evt=None, reset=lambda: None, slot=slot)
        slots_used.append(slot)
        return cap._plans[key]

    def enc_into(cp, n, dec, row0, stream=None):
        x = cp.x_in.t[:n, :, :, :3].double()
        key = (x.sum((1, 2, 3)) * 977.0 + x[:, ::7, ::5].sum((1, 2, 3)) * 131.0).abs()
        dec.fp[row0:row0 + n] = torch.stack([torch.zeros(n), 100 + (key.long() % 40000), torch.full((n,), 2.0)], 1).int()

    def dec_merged(dec, n, max_new, stream=None):
        out = dec.fp[:n].clone()
        dec.fp.fill_(7)                      # whoever reads this plan after its decode sees garbage
        return out
    monkeypatch.setattr(cap, "decode_plans", fake_plans)
    monkeypatch.setattr(cap, "_encode_into", enc_into)
    monkeypatch.setattr(cap, "_decode_merged", dec_merged)
    sp = ScreenParser(det, cap, box_threshold=0.5, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=320, batch_size=4)
    sp.max_new_tokens = 1                       # the one-frame batch fits one micro-batch and takes the real decode plan: one step of it
    assert sp.device_glue
    batches = [([torch.from_numpy(synthetic_screenshot(s, 640, 480)) for s in seeds], [synthetic_ocr(s, 640, 480, 10) for s in seeds])
               for seeds in ((0, 1), (2,), (3, 0))]
    want, crops_want = [], []
    for f, o in batches:
        want.append(sp.parse_batch(f, o, return_ids=True, pad_to=2))
        crops_want.append(sp.last_crops)
    assert set(slots_used) == {0}
    del slots_used[:]
    got, crops_got = [], []
    for res in sp.parse_stream(iter(batches), return_ids=True, pad_to=2):
        got.append(res)
        crops_got.append(sp.last_crops)
    assert len(got) == 3 and crops_got == crops_want
    assert len(slots_used) >= 2 and all(a != b for a, b in zip(slots_used, slots_used[1:])), slots_used   # merged batches alternate plans
    for (el_w, ids_w), (el_g, ids_g) in zip(want, got):
        assert el_g == el_w
        assert [[r.tolist() for r in f] for f in ids_g] == [[r.tolist() for r in f] for f in ids_w]
    assert len({tuple(r.tolist()) for _, ids in want for f in ids for r in f}) >= 10


def test_merged_decode_equals_per_micro_batch_decode(emu, monkeypatch):
    """ScreenParser.caption over more crops than one micro-batch holds: encode per micro-batch, cross-attention K / V copied into ONE
    decode plan, 20 (here 2) steps over all rows at once (florence.py::_DecodePlans) — the same ids as decoding every micro-batch on its
    own (OMNI_MERGED_DECODE=0)."""
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.pipeline import ScreenParser
    from omniparser_amd.synth import synthetic_screenshot
    from tools.make_weights import ensure_caption_checkpoint
    from conftest import small_vocab_caption_checkpoint        # 8192-row token table: see its docstring
    cap = Florence2Captioner(small_vocab_caption_checkpoint(0), "cuda", precision="f32", resolution=64)
    monkeypatch.setattr(Florence2Captioner, "decode_bucket", staticmethod(lambda n: 8))       # a 128-row lm_head step costs the emulation minutes
    import omniparser_amd.florence as FL
    monkeypatch.setattr(FL, "_BUCKETS", (2, 128))                                            # 2-row encode plans for the 2-crop micro-batches
    frame = torch.from_numpy(synthetic_screenshot(3, 640, 480))
    rects = [[[10, 20, 60, 70], [300, 200, 340, 260], [500, 100, 620, 140]], [[40, 40, 90, 80], [200, 300, 280, 360]]]
    got = {}
    # the decode plan has 8 rows for 5 crops: its rows 5..7 are decoded although nobody reads them.  Poison them (and nothing else) with
    # NaN, as recycled allocator memory can: all-NaN logits must not turn into an out-of-range token id for the next embedding gather
    # (the one-process GPU suite of round 3 died on exactly that), and rows 0..4 must not notice
    real_plans = cap.decode_plans

    def poisoned(*a, **k):
        dec = real_plans(*a, **k)
        for kv in dec.cross_kv:
            kv.t[5:] = float("nan")
        return dec
    monkeypatch.setattr(cap, "decode_plans", poisoned)
    for mode in ("1", "0"):
        monkeypatch.setenv("OMNI_MERGED_DECODE", mode)
        sp = ScreenParser(None, cap, batch_size=2)
        sp.max_new_tokens = 2
        out = sp.caption([frame, frame], rects)
        got[mode] = [[row.tolist() for _, row in f] for f in out]
    assert any(k[0] == "dec" for k in cap._plans)                                          # the merged path ran
    # round 6: the 1-crop remainder of the merged batch ran as an exact-row twin in the buffers of the 2-row plan set, not as a bucket plan
    strip = lambda rows: [[t for t in r if t != cap.w.pad] for r in rows]
    # (the count is new the first time: the ladder capacity above it is the full plan; a repeated count gets its exact twin)
    assert cap.exact_rows and getattr(cap, "row_graph_builds", 0) == 0
    monkeypatch.setenv("OMNI_MERGED_DECODE", "1")
    sp = ScreenParser(None, cap, batch_size=2)
    sp.max_new_tokens = 2
    again = [[row.tolist() for _, row in f] for f in sp.caption([frame, frame], rects)]
    assert cap.row_graph_builds == 1 and sorted(cap._plans[(2, 64, 2)]._row_plans) == [1]
    assert [strip(f) for f in again] == [strip(f) for f in got["0"]]
    dec = next(v for k, v in cap._plans.items() if k[0] == "dec")
    assert int(dec.ids.min()) >= 0 and int(dec.ids.max()) < cap.w.vocab, "a padding row produced an out-of-range token id"
    assert [len(f) for f in got["1"]] == [3, 2]
    assert [strip(f) for f in got["1"]] == [strip(f) for f in got["0"]]


def test_exact_row_encode_twin_in_the_full_plans_buffers(emu):
    """florence.py::_CaptionPlans.encode_rows on the emulation: 3 rows of a 4-row plan set's buffers — same features / encoder output /
    cross-attention K and V as the full plan on those rows, the full plan unharmed afterwards, the twin cached."""
    import gpu_checks as G
    out, cap = G.check_exact_rows(R=64, n=3, capacity=4, small_vocab=True)
    print(out)
    assert out["twin_cached"] and out["builds"] == 1 and out["twin_ops"] == out["full_ops"]
    assert abs(out["twin_gflop_per_crop"] - out["full_gflop_per_crop"]) < 1e-6 * out["full_gflop_per_crop"]
    for k, v in out.items():
        if isinstance(v, dict):
            assert v["rel"] <= 1e-5 and v["full_again_bitwise"], (k, out)
