"""Batched screen parser: B screenshots -> parsed elements, everything model-side on device.

The reference parses one screenshot at a time (ref:util/omniparser.py:16-32).  Screenshots are
independent units, so here a batch is parsed as: ONE detector graph over all B frames -> host glue per
frame (the reference's list semantics, util/utils.py) -> the crops of ALL frames packed into 128-crop
caption micro-batches (ref batch_size=128) -> on-device greedy decode.  Results equal a per-frame
`get_som_labeled_img(...)[2]` call (same functions underneath); only the packing differs.
"""
import contextlib
import os
from types import SimpleNamespace
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from .florence import CLIP_MEAN, CLIP_STD, Florence2Captioner
from .util import utils as U
from .util.yolov9 import YOLOv9Detector


OCR_CAP = 1024           # OMNI_OP_GLUE capacities (csrc/glue_ops.hip)
MASK_WORDS = OCR_CAP // 64


def _f64_bits(x: float):
    import struct
    lo, hi = struct.unpack("<ii", struct.pack("<d", float(x)))
    return lo, hi


class _GlueState:
    """Device side of the detect -> caption hand-off for one detector plan (batch B): fixed buffers + one small plan of B
    OMNI_OP_GLUE ops replayed right behind the detector graph on the detector's stream."""

    def __init__(self, dp, det, iw, ih, thr):
        dev, B, md = det.device, dp.batch, dp.out_boxes.shape[1]
        self.B, self.md = B, md
        with torch.cuda.stream(det.stream):
            self.ocr = torch.zeros(B, OCR_CAP, 4, dtype=torch.float64, device=dev)
            self.meta = torch.zeros(B, 2 + 2 * OCR_CAP, dtype=torch.int32, device=dev)
            self.elems = torch.zeros(B, md + OCR_CAP, 2, dtype=torch.int32, device=dev)
            self.crops = torch.zeros(B, md, 4, dtype=torch.int32, device=dev)
            self.counts = torch.zeros(B, 4, dtype=torch.int32, device=dev)
            self.donors = torch.zeros(B, md, MASK_WORDS, dtype=torch.int64, device=dev)
        self.h_ocr = torch.zeros(B, OCR_CAP, 4, dtype=torch.float64)          # host staging of the per-frame OCR tables (a few KB)
        self.h_meta = torch.zeros(B, 2 + 2 * OCR_CAP, dtype=torch.int32)
        lo, hi = _f64_bits(thr)
        ops = [L.make_op(L.OP_GLUE, L.F32,
                         p=[dp.out_boxes[b].data_ptr(), dp.out_count[b:].data_ptr(), self.ocr[b].data_ptr(), self.meta[b].data_ptr(),
                            self.elems[b].data_ptr(), self.crops[b].data_ptr(), self.counts[b].data_ptr(), self.donors[b].data_ptr()],
                         i={0: md, 1: OCR_CAP, 2: iw, 3: ih, 4: MASK_WORDS, 5: 0, 6: md + OCR_CAP, 7: 1, 8: lo, 9: hi})
               for b in range(B)]
        # detector ops AND hand-off ops as ONE plan, captured as ONE hipGraph (kernel nodes only): one launch per batch.  Round 2
        # saw the second replay of the detector graph stall whenever hand-off kernels followed it; the detector graph then held a
        # hipMemsetAsync node (the candidate-counter reset of OMNI_OP_DETECT_DECODE).  With that reset done by a kernel both
        # arrangements ran 100 consecutive replays on the MI355X with the host twin's results (profiles/r3_s1_handoff_graph_replays.jsonl)
        self.plan = L.Plan(list(dp.plan.ops) + ops)
        det.stream.synchronize()
        if det.use_graph:
            self.plan.run(det.stream)
            det.stream.synchronize()
            self.plan.capture(det.stream)
            det.stream.synchronize()

    def launch(self, det):
        if det.use_graph:
            self.plan.replay(det.stream)
        else:
            self.plan.run(det.stream)


class ScreenParser:
    def __init__(self, detector: YOLOv9Detector, captioner: Florence2Captioner, processor=None,
                 box_threshold=0.05, iou_threshold=0.7, nms_iou=0.1, max_det=300, imgsz=640, batch_size=128,
                 tile_large=False):
        self.det, self.cap = detector, captioner
        self.tile_large = tile_large      # False = reference behaviour (whole frame letterboxed to `imgsz`)
        # detect -> caption hand-off ON THE DEVICE (csrc/glue_ops.hip, inside the detector's graph): boxes never visit the host between
        # the stages.  OMNI_DEVICE_GLUE=0 = the host twin (`glue`: the reference's list code), also taken automatically for inputs
        # beyond the kernel's capacities (more than 512 detections or 1024 OCR boxes per frame)
        self.device_glue = os.environ.get("OMNI_DEVICE_GLUE", "1") != "0" and max_det <= 512
        self.proc = processor or (U.FlorenceProcessor(captioner.w.dir) if captioner is not None else None)
        self.box_threshold, self.iou_threshold, self.nms_iou = box_threshold, iou_threshold, nms_iou
        self.max_det, self.imgsz, self.batch_size = max_det, imgsz, max(1, min(int(batch_size), 128))   # caption plan capacity
        self.max_new_tokens = 20          # ref:util/utils.py:125 generate(max_new_tokens=20)
        self.encode_lanes = 2             # parse_stream: caption micro-batches in flight at once (HIP streams; 1 = one after the other)
        self._ev = {}                     # HIP events of the last parse_batch (stage boundaries on the streams the stages run on)
        self.stats = {}

    # ---- stage 1: detector over the whole batch (one graph launch)
    @torch.inference_mode()
    def detect(self, frames: Sequence[torch.Tensor], pad_to: Optional[int] = None):
        """`pad_to`: run the plan of that batch size even for fewer frames (the unused slots keep whatever they held
        and their results are ignored) — streams with ragged batches then need ONE plan per resolution."""
        ih, iw = frames[0].shape[:2]
        dp = self.det.get_plan(iw, ih, self.imgsz, self.box_threshold, self.nms_iou, self.max_det, batch=max(len(frames), pad_to or 0))
        self.det.stream.wait_stream(torch.cuda.current_stream(frames[0].device))     # frames the caller is still producing on its stream
        with torch.cuda.stream(self.det.stream):
            for bi, f in enumerate(frames):
                dp.img[bi].copy_(f, non_blocking=True)
            dp.launch(self.det)
            counts = dp.out_count.cpu()                     # sync
            boxes = dp.out_boxes.cpu()
        return [boxes[bi, : int(counts[bi])] for bi in range(len(frames))]

    # ---- stage 1b: tiled detection for frames larger than 1080p (BASELINE configs[4]; policy is OURS, the
    #      reference letterboxes the whole image: SURVEY 0.7).  Tiles overlap by 64 px, each tile goes through the
    #      ordinary detector (one batch), boxes are shifted back and a global batched_nms(nms_iou)[:max_det] + clamp
    #      merges them (same NMS kernel as inside the detector).  oracle/tiling_ref.py is the CPU statement.
    @staticmethod
    def tile_origins(iw, ih, tile_w=1952, tile_h=1112, overlap=64):
        def axis(n, t):
            if n <= t:
                return [0]
            k = -(-(n - overlap) // (t - overlap))
            step = (n - t) / (k - 1)
            return [int(round(i * step)) for i in range(k)]
        return [(x, y) for y in axis(ih, tile_h) for x in axis(iw, tile_w)], min(tile_w, iw), min(tile_h, ih)

    @torch.inference_mode()
    def detect_tiled(self, frame: torch.Tensor):
        ih, iw = frame.shape[:2]
        origins, tw, th = self.tile_origins(iw, ih)
        dp = self.det.get_plan(tw, th, self.imgsz, self.box_threshold, self.nms_iou, self.max_det, batch=len(origins))
        # The detector's stream is a NON-BLOCKING stream: nothing orders it behind work the caller queued on its own (current) stream.
        # Rounds 2-5 cut the tiles with `.contiguous()` on the CALLER's stream and copied them into the plan on the detector's — a
        # race a few microseconds wide that the round-6 closing check lost once in six suite runs (two `detect_tiled` calls on the same
        # frame returned different boxes).  Now the detector's stream first waits for everything the caller has queued (the frame itself
        # may still be in flight, e.g. a service's upload), and the tiles are cut ON it.
        self.det.stream.wait_stream(torch.cuda.current_stream(frame.device))
        with torch.cuda.stream(self.det.stream):
            tiles = [frame[y:y + th, x:x + tw].contiguous() for (x, y) in origins]
            for bi, t in enumerate(tiles):
                dp.img[bi].copy_(t, non_blocking=True)
            dp.launch(self.det)
            counts = dp.out_count.cpu()
            boxes, scores, cls = dp.out_boxes.cpu(), dp.out_scores.cpu(), dp.out_cls.cpu()
        recs = []
        for bi, (x0, y0) in enumerate(origins):
            k = int(counts[bi])
            b = boxes[bi, :k].numpy() + np.asarray([x0, y0, x0, y0], dtype=np.float32)
            r = np.zeros((k, 8), dtype=np.float32)
            r[:, :4] = b
            r[:, 4] = scores[bi, :k].numpy()
            ri = r.view(np.int32)
            ri[:, 5] = cls[bi, :k].numpy()
            recs.append(r)
        rec = np.concatenate(recs, 0) if recs else np.zeros((0, 8), dtype=np.float32)
        n = rec.shape[0]
        rec.view(np.int32)[:, 6] = np.arange(n)               # stable tie-break = concatenation order (tile-major)
        dev = self.det.device
        cap_n = max(n, 1)
        with torch.cuda.stream(self.det.stream):               # scratch is created AND consumed on the detector's stream
            cand = torch.from_numpy(np.concatenate([rec, np.zeros((cap_n - n, 8), dtype=np.float32)], 0)).to(dev)
            count = torch.tensor([n], dtype=torch.int32, device=dev)
            srt = torch.zeros((cap_n + 1) * 8, dtype=torch.float32, device=dev)
            mask = torch.empty(cap_n * ((cap_n + 63) // 64), dtype=torch.int64, device=dev)
            ob = torch.zeros(self.max_det, 4, device=dev); osc = torch.zeros(self.max_det, device=dev)
            oc = torch.zeros(self.max_det, dtype=torch.int32, device=dev); on = torch.zeros(1, dtype=torch.int32, device=dev)
            op = L.make_op(L.OP_NMS, L.F32, p=[cand.data_ptr(), count.data_ptr(), srt.data_ptr(), mask.data_ptr(), ob.data_ptr(),
                                              osc.data_ptr(), oc.data_ptr(), on.data_ptr()],
                           i={0: cap_n, 1: self.max_det, 2: iw, 3: ih}, f={0: self.nms_iou})
            L.launch(op, self.det.stream)
            k = int(on.cpu())
            return ob[:k].cpu(), osc[:k].cpu(), oc[:k].cpu().long()

    # ---- stage 2: host glue (reference semantics) -> elements + crop boxes
    def glue(self, xyxy_px: torch.Tensor, w: int, h: int, ocr_bbox, ocr_text):
        xyxy = xyxy_px / torch.Tensor([w, h, w, h])
        if ocr_bbox:
            ocr_r = (torch.tensor(ocr_bbox) / torch.Tensor([w, h, w, h])).tolist()
        else:
            ocr_r, ocr_text = [], []
        ocr_el = [{"type": "text", "bbox": b, "interactivity": False, "content": t, "source": "box_ocr_content_ocr"}
                  for b, t in zip(ocr_r, ocr_text) if U.int_box_area(b, w, h) > 0]
        icon_el = [{"type": "icon", "bbox": b, "interactivity": True, "content": None}
                   for b in xyxy.tolist() if U.int_box_area(b, w, h) > 0]
        filtered = U.remove_overlap_new(boxes=icon_el, iou_threshold=self.iou_threshold, ocr_bbox=ocr_el)
        if not ocr_el:
            filtered = [e if isinstance(e, dict) else {"type": "icon", "bbox": e, "interactivity": True, "content": None,
                                                        "source": "box_yolo_content_yolo"} for e in filtered]
        elems = sorted(filtered, key=lambda x: x["content"] is None)
        start = next((i for i, b in enumerate(elems) if b["content"] is None), -1)
        boxes = [e["bbox"] for e in elems]
        non_ocr = boxes[start:] if start else boxes        # ref:util/utils.py:92-95 quirk kept
        return elems, U.crop_boxes_px(non_ocr, w, h)

    # ---- stage 2': the same hand-off on the device.  Host work before the launch touches the OCR list only (its ratio
    #      boxes, int_box_area filter and dict-equality classes do not depend on the detector); after the launch the host
    #      reads four counters per frame, and builds the element dicts at the very end from the device tables.
    @staticmethod
    def ocr_elements(w, h, ocr_bbox, ocr_text):
        if ocr_bbox:
            ocr_r = (torch.tensor(ocr_bbox) / torch.Tensor([w, h, w, h])).tolist()
        else:
            ocr_r, ocr_text = [], []
        return [{"type": "text", "bbox": b, "interactivity": False, "content": t, "source": "box_ocr_content_ocr"}
                for b, t in zip(ocr_r, ocr_text) if U.int_box_area(b, w, h) > 0]

    @torch.inference_mode()
    def detect_glue(self, frames, ocr, pad_to=None, snapshot=False):
        """snapshot: the tables the rest of the batch reads (detector outputs, element table, donor masks, crop rectangles: ~0.5 MB)
        are copied behind the graph on the detector's stream and returned instead of the plan's own buffers, so the NEXT batch's
        detector pass may overwrite those while this batch is still being captioned (parse_stream)."""
        ih, iw = frames[0].shape[:2]
        det = self.det
        dp = det.get_plan(iw, ih, self.imgsz, self.box_threshold, self.nms_iou, self.max_det, batch=max(len(frames), pad_to or 0))
        ocr_els = [self.ocr_elements(iw, ih, *(reversed(ocr[fi]) if ocr is not None else ([], []))) for fi in range(len(frames))]
        if any(len(els) > OCR_CAP for els in ocr_els):
            return None                                   # beyond the kernel's OCR capacity: the caller takes the host twin
        key = ("glue", float(self.iou_threshold))
        gs = getattr(dp, "_glue", {}).get(key)
        if gs is None:
            with torch.cuda.device(det.device):
                gs = _GlueState(dp, det, iw, ih, self.iou_threshold)
            dp._glue = {**getattr(dp, "_glue", {}), key: gs}
        gs.h_meta.zero_()
        for fi in range(len(frames)):
            els = ocr_els[fi]
            m = len(els)
            if m:
                gs.h_ocr[fi, :m] = torch.tensor([e["bbox"] for e in els], dtype=torch.float64)
                first, seen = {}, {}
                tab = gs.h_meta[fi, 2:2 + 2 * m].view(m, 2)
                for j, e in enumerate(els):                      # list.remove() works on dict equality: bbox + content decide it
                    k = (tuple(e["bbox"]), e["content"])
                    cls = first.setdefault(k, j)
                    tab[j, 0] = cls
                    tab[j, 1] = seen.get(cls, 0)
                    seen[cls] = seen.get(cls, 0) + 1
            gs.h_meta[fi, 0] = m
        det.stream.wait_stream(torch.cuda.current_stream(frames[0].device))          # frames the caller is still producing on its stream
        with torch.cuda.stream(det.stream):
            self._ev["det0"] = det.stream.record_event(torch.cuda.Event(enable_timing=True))
            for bi, f in enumerate(frames):
                dp.img[bi].copy_(f, non_blocking=True)
            gs.ocr.copy_(gs.h_ocr)
            gs.meta.copy_(gs.h_meta)
            gs.launch(det)                                        # detector + hand-off: one graph
            self._ev["det1"] = det.stream.record_event(torch.cuda.Event(enable_timing=True))
            tables = (dp, gs)
            if snapshot:
                snap = SimpleNamespace(out_boxes=dp.out_boxes.clone(), out_count=dp.out_count.clone(), elems=gs.elems.clone(),
                                       donors=gs.donors.clone(), crops=gs.crops.clone())
                tables = (snap, snap)
            counts = gs.counts.to("cpu", copy=True)               # the one synchronising read-back: 4 ints per frame
        return tables[0], tables[1], ocr_els, counts

    @torch.inference_mode()
    def assemble(self, dp, gs, ocr_els, counts, iw, ih, n_frames):
        """element dicts of every frame from the device tables (called once, after the caption micro-batches were queued)."""
        with torch.cuda.stream(self.det.stream):
            boxes = dp.out_boxes.cpu(); kcnt = dp.out_count.cpu(); elems = gs.elems.cpu(); donors = gs.donors.cpu()
        out = []
        for f in range(n_frames):
            k = int(kcnt[f])
            ratios = (boxes[f, :k] / torch.Tensor([iw, ih, iw, ih])).tolist()
            els = []
            for kind, src in elems[f, : int(counts[f, 0])].tolist():
                if kind == 0:
                    els.append(ocr_els[f][src])
                    continue
                label = None
                if kind == 1:
                    label = ""
                    for wd in range(MASK_WORDS):
                        bits = int(donors[f, src, wd]) & 0xFFFFFFFFFFFFFFFF
                        while bits:
                            low = bits & -bits
                            label += ocr_els[f][wd * 64 + low.bit_length() - 1]["content"] + " "
                            bits ^= low
                els.append({"type": "icon", "bbox": ratios[src], "interactivity": True, "content": label,
                            "source": "box_yolo_content_ocr" if kind == 1 else "box_yolo_content_yolo"})
            out.append(els)
        return out

    # ---- stage 3: caption all crops of all frames in packed micro-batches
    @torch.inference_mode()
    def caption(self, frames: Sequence[torch.Tensor], crops_per_frame, max_new_tokens=None, crops_dev: Optional[torch.Tensor] = None):
        """crops_per_frame: host rectangles per frame, or — with `crops_dev` (int32 [frames, max_det, 4] on the device, rows in
        caption order) — just the number of crops per frame: the rectangles then never visit the host."""
        return self.caption_finish(self.caption_launch(frames, crops_per_frame, max_new_tokens, crops_dev))

    @torch.inference_mode()
    def caption_launch(self, frames: Sequence[torch.Tensor], crops_per_frame, max_new_tokens=None, crops_dev: Optional[torch.Tensor] = None,
                       overlap=False):
        """queue the crop / encode / decode work of every micro-batch on the captioner's stream; nothing is read back.  The handle
        keeps the frames alive until `caption_finish`.  overlap (parse_stream): micro-batches alternate between `self.encode_lanes`
        HIP streams, and the merged decode runs on a further stream on one of two alternating decode plans, so the next batch's
        encode does not wait for it."""
        cap = self.cap
        R = cap.resolution
        max_new_tokens = max_new_tokens or self.max_new_tokens
        if crops_dev is not None:
            flat = [(fi, k) for fi, n in enumerate(crops_per_frame) for k in range(int(n))]
        else:
            flat = [(fi, b) for fi, cl in enumerate(crops_per_frame) for b in cl]
        ids_all = []
        if cap._lut is None:
            cap._lut = torch.from_numpy((np.arange(256).astype(np.float64) * (1 / 255)).astype(np.float32)).to(cap.device)
            if R != 64:
                b, k = L.resample_coeffs(64, R, 1)
                cap._bic = (torch.from_numpy(b).to(cap.device), torch.from_numpy(k).to(cap.device), k.shape[1])
        esz = 4 if cap.dtype == L.F32 else 2
        # more than one micro-batch: the encode side runs per micro-batch, the 20 decode steps ONCE over all crops
        # (florence.py::_DecodePlans); OMNI_MERGED_DECODE=0 = every micro-batch decodes on its own (A/B knob)
        merged = len(flat) > self.batch_size and os.environ.get("OMNI_MERGED_DECODE", "1") != "0"
        # overlap (parse_stream): micro-batches alternate between two encode lanes (HIP streams); 128-row plans exist once per lane, the
        # smaller capacities once — a plan's `free_evt` orders its next use, on whichever lane, behind its last one
        lanes = [cap.encode_lane(k) for k in range(max(1, self.encode_lanes))] if overlap else [cap.stream]
        cap.begin_batch()              # the plan sets this batch takes from the captioner's cache stay resident until the next batch
        if overlap and merged:
            self._dec_slot = 1 - getattr(self, "_dec_slot", 1)
        dec = cap.decode_plans(cap.decode_bucket(len(flat)), R, max_new_tokens, slot=self._dec_slot if overlap else 0) if merged else None
        if merged and not overlap:
            with torch.cuda.stream(cap.stream):
                dec.reset()
        if not overlap:
            self._ev["cap0"] = cap.stream.record_event(torch.cuda.Event(enable_timing=True))
        used = []
        for mbi, s in enumerate(range(0, len(flat), self.batch_size)):
            chunk = flat[s:s + self.batch_size]
            n = len(chunk)
            lane = 0
            if overlap and merged:
                lane = self._mb_count = (getattr(self, "_mb_count", -1) + 1) % len(lanes)
            stream = lanes[lane]
            # full micro-batches: one plan set per lane.  The remainder of a merged batch encodes exactly its n rows as a second graph over
            # the buffers of ITS LANE's full-capacity plan set (florence.py::_CaptionPlans.encode_rows, taken inside `_encode_into`);
            # otherwise (single micro-batch, OMNI_EXACT_ROWS=0) a padded bucket plan with buffers of its own
            if merged and cap.exact_rows and n < self.batch_size:
                cp = cap.plans(cap.bucket(self.batch_size), R, max_new_tokens, slot=lane)
            else:
                cp = cap.plans(cap.bucket(n), R, max_new_tokens, slot=lane if n == self.batch_size else 0)
            with torch.cuda.stream(stream):
                if cp.free_evt is not None:
                    stream.wait_event(cp.free_evt)
                if merged and dec.free_evt is not None and stream not in used:
                    stream.wait_event(dec.free_evt)      # the decode two batches ago read this decode plan's cross-attention K/V
                if not merged:
                    cp.reset()
                c64 = torch.empty((n, 64, 64, 3), dtype=torch.uint8, device=cap.device)
                tmp = torch.empty((n, 64, R, 3), dtype=torch.uint8, device=cap.device) if R != 64 else None
                bb, kk, ks = cap._bic if R != 64 else (None, None, 0)
                if crops_dev is None:
                    bx = torch.tensor([b for _, b in chunk], dtype=torch.int32).to(cap.device, non_blocking=True)
                o = 0
                while o < n:                                   # one crop launch per source frame run
                    fi = chunk[o][0]
                    e = o
                    while e < n and chunk[e][0] == fi:
                        e += 1
                    H, W = frames[fi].shape[:2]
                    rects = crops_dev[fi, chunk[o][1]:] if crops_dev is not None else bx[o:]     # rows of one frame are contiguous
                    op = L.make_op(
                        L.OP_CROP_RESIZE, cap.dtype,
                        p=[frames[fi].data_ptr(), rects.data_ptr(), c64[o:].data_ptr(), tmp[o:].data_ptr() if tmp is not None else None,
                           cp.x_in.ptr + o * R * R * cp.x_in.ld * esz, bb.data_ptr() if bb is not None else None,
                           kk.data_ptr() if kk is not None else None, cap._lut.data_ptr()],
                        i={0: e - o, 1: H, 2: W, 3: R, 4: ks, 13: cp.x_in.ld},
                        f={0: CLIP_MEAN[0], 1: CLIP_MEAN[1], 2: CLIP_MEAN[2], 3: CLIP_STD[0], 4: CLIP_STD[1], 5: CLIP_STD[2]})
                    L.launch(op, stream)
                    o = e
                if merged:
                    cap._encode_into(cp, n, dec, s, stream)
                else:
                    ids_all.append(cap._run(cp, n, max_new_tokens, defer=True))   # keep the GPU fed: no sync between micro-batches
                if overlap:
                    cp.free_evt = stream.record_event()
            if stream not in used:
                used.append(stream)
        ids_stream = cap.stream
        if merged and not overlap:
            with torch.cuda.stream(cap.stream):
                self._ev["cap1"] = cap.stream.record_event(torch.cuda.Event(enable_timing=True))
                ids_all.append(cap._decode_merged(dec, len(flat), max_new_tokens))
                self._ev["cap2"] = cap.stream.record_event(torch.cuda.Event(enable_timing=True))
        elif merged:
            ids_stream = cap.dec_stream
            encoded = [st.record_event() for st in used]
            with torch.cuda.stream(ids_stream):
                for ev in encoded:
                    ids_stream.wait_event(ev)
                dec.reset()
                ids_all.append(cap._decode_merged(dec, len(flat), max_new_tokens, ids_stream))
                dec.free_evt = ids_stream.record_event()
        if not overlap:
            self._ev["capE"] = cap.stream.record_event(torch.cuda.Event(enable_timing=True))
        return (list(frames), flat, ids_all, ids_stream)

    def _settle_gc(self):
        """Python's generational collector, not the GPU, produced the one-step-in-twenty +40...70 ms of the round-5 bench lines
        (profiles/r6_s2_gc_*.json: one gen-2 collection inside the driver's 20 steps = the one 733 ms step; none with the collector off
        or frozen).  The plans of a parser are tens of thousands of long-lived Python objects (op descriptors, views, tensors): every
        full collection walks them all while the GPU pipeline waits for the host to queue the next batch.  After each of the first
        batches (all plan sets of the steady state exist by then: two encode lanes, the remainder plan, two decode plans) the survivors
        are moved to the permanent generation (`gc.freeze`), so later collections only look at what a batch allocates.  The collector
        stays on; OMNI_GC_FREEZE=0 turns this off."""
        n = getattr(self, "_gc_settled", 0)
        if n >= 3 or os.environ.get("OMNI_GC_FREEZE", "1") == "0":
            return
        import gc
        gc.unfreeze()                  # what an earlier parser of this process froze and has since died must be collectable (its plans hold HBM)
        gc.collect()
        gc.freeze()
        self._gc_settled = n + 1

    def _check_range(self):
        """Range guard of the split-f16 GEMM operands (include/omni_amd.h::omni_overflow_count), read where the host has just waited for
        a batch's ids anyway: a value beyond the f16 range was clamped (format B) or lost (format A) somewhere in the last batch(es) —
        the fp32 reference has no such limit, so the result may differ from it.  Accumulated in `range_overflow_total`, the last
        reading in `stats["split_overflow"]` (filled by the callers), fatal with OMNI_STRICT_RANGE=1 (what the parity tests run with)."""
        n = L.overflow_count(reset=True)
        self.range_overflow_last = n
        self.range_overflow_total = getattr(self, "range_overflow_total", 0) + n
        if n and os.environ.get("OMNI_STRICT_RANGE", "0") == "1":
            raise L.OmniError(f"split-f16 range guard: {n} threads produced a GEMM operand beyond +-65504 in this batch "
                              "(the fp32 reference has no such limit; results may differ from it)")

    @torch.inference_mode()
    def caption_finish(self, handle):
        frames, flat, ids_all, ids_stream = handle
        cap = self.cap
        # single read-back point.  The snapshots were produced on one of the captioner's streams (non-blocking streams: the default
        # stream does not order against them), so the copies are issued ON that stream — stream order alone makes them see the finished ids
        with torch.cuda.stream(ids_stream):
            ids_all = [t.cpu() for t in ids_all]
        ids_all = [cap._finish_ids(t.long()) for t in ids_all]
        self._check_range()
        self._settle_gc()
        out = [[] for _ in frames]
        k = 0
        for ids in ids_all:
            texts = [t.strip() for t in self.proc.batch_decode(ids, skip_special_tokens=True)]
            for t, row in zip(texts, ids):
                out[flat[k][0]].append((t, row))
                k += 1
        return out

    @torch.inference_mode()
    def parse_batch(self, frames: Sequence[torch.Tensor], ocr: Optional[Sequence] = None, return_ids=False,
                    pad_to: Optional[int] = None):
        """frames: uint8 [H,W,3] device tensors (same size); ocr: per frame (texts, xyxy px boxes) or None;
        pad_to: detector plan batch size to use when fewer frames arrive (see `detect`)."""
        ih, iw = frames[0].shape[:2]
        with self.det._lock, self.cap._lock:
            return self._parse_batch_locked(frames, ocr, return_ids, iw, ih, pad_to)

    def parse_stream(self, batches, return_ids=False, pad_to: Optional[int] = None):
        """Generator over an iterable of (frames, ocr) batches -> what `parse_batch` returns for each, in order, as a software
        pipeline: the detector pass and the host hand-off of batch i+1 run on a helper thread (their stream is the detector's)
        while the captions of batch i occupy the GPU, and the caption work of batch i+1 is queued BEFORE the read-back of batch i
        blocks — the GPU never waits for the host between batches.  Same kernels, same order per batch, same results as
        `parse_batch`.  (Host hand-off only: with OMNI_DEVICE_GLUE the crop table of a batch lives in per-plan device buffers.)"""
        from concurrent.futures import ThreadPoolExecutor
        if self.device_glue:
            yield from self._parse_stream_device(batches, return_ids, pad_to)
            return

        def stage_a(frames, ocr):
            with torch.inference_mode(), self.det._lock:
                ih, iw = frames[0].shape[:2]
                det_boxes = self.detect(frames, pad_to)
                elems_all, crops_all = [], []
                for fi, xy in enumerate(det_boxes):
                    texts, boxes = ocr[fi] if ocr is not None else ([], [])
                    el, cr = self.glue(xy, iw, ih, boxes, texts)
                    elems_all.append(el); crops_all.append(cr)
                return frames, elems_all, crops_all, [len(b) for b in det_boxes]

        def finish(pending):
            handle, elems_all, crops_all, nbox = pending
            with torch.inference_mode(), self.cap._lock:
                caps = self.caption_finish(handle)
            ids_out = self._fill_captions(elems_all, caps)
            self.stats = {"crops": [len(c) for c in crops_all], "boxes": nbox, "split_overflow": getattr(self, "range_overflow_last", 0)}
            self.last_crops = crops_all
            return (elems_all, ids_out) if return_ids else elems_all

        it = iter(batches)
        first = next(it, None)
        if first is None:
            return
        with ThreadPoolExecutor(1) as helper:
            fut = helper.submit(stage_a, *first)
            pending = None
            while fut is not None:
                frames, elems_all, crops_all, nbox = fut.result()
                nxt = next(it, None)
                fut = helper.submit(stage_a, *nxt) if nxt is not None else None
                with torch.inference_mode(), self.cap._lock:
                    handle = self.caption_launch(frames, crops_all)
                if pending is not None:
                    yield finish(pending)
                pending = (handle, elems_all, crops_all, nbox)
            yield finish(pending)

    def _parse_stream_device(self, batches, return_ids, pad_to):
        """parse_stream with the device hand-off (the default): four HIP streams, one host thread.  Batch i+1's detector + hand-off
        graph runs on the detector's stream while batch i encodes; its tables are snapshotted (`detect_glue`) so nothing of batch i
        reads the detector plan's buffers afterwards; caption micro-batches alternate between two encode streams (the HBM-bound
        kernels of one fill what the MFMA-bound GEMMs of the other leave idle); batch i's 20 decode steps run on a fourth stream, on
        one of two decode plans, while batch i+1 encodes.  Same kernels on the same data per batch as `parse_batch`."""
        def finish(p):
            handle, snap, ocr_els, counts, iw, ih, n_frames, n_crops = p
            with torch.inference_mode(), self.cap._lock:
                caps = self.caption_finish(handle)
                elems_all = self.assemble(snap, snap, ocr_els, counts, iw, ih, n_frames)
            ids_out = self._fill_captions(elems_all, caps)
            self.stats = {"crops": n_crops, "boxes": [int(v) for v in snap.out_count[:n_frames].tolist()], "split_overflow": getattr(self, "range_overflow_last", 0)}
            self.last_crops = [snap.crops[f, :n].tolist() for f, n in enumerate(n_crops)]
            return (elems_all, ids_out) if return_ids else elems_all

        try:
            yield from self._stream_loop(batches, return_ids, pad_to, finish)
        finally:
            torch.cuda.synchronize(self.cap.device)      # an abandoned generator leaves no work behind on the side streams

    def _stream_loop(self, batches, return_ids, pad_to, finish):
        pending = None
        for frames, ocr in batches:
            ih, iw = frames[0].shape[:2]
            tiled = self.tile_large and (iw > 1952 or ih > 1112)
            with self.det._lock:
                handed = None if tiled else self.detect_glue(frames, ocr, pad_to, snapshot=True)
            if handed is None:                       # beyond the hand-off kernel's capacities / tiled: drain, then the ordinary path
                if pending is not None:
                    yield finish(pending)
                    pending = None
                yield self.parse_batch(frames, ocr, return_ids=return_ids, pad_to=pad_to)
                continue
            snap, _, ocr_els, counts = handed
            n_crops = [int(counts[f, 1]) for f in range(len(frames))]
            with torch.inference_mode(), self.cap._lock:
                handle = self.caption_launch(frames, n_crops, crops_dev=snap.crops, overlap=True)
            if pending is not None:
                yield finish(pending)
            pending = (handle, snap, ocr_els, counts, iw, ih, len(frames), n_crops)
        if pending is not None:
            yield finish(pending)

    def stage_ms(self):
        """device time of the stages of the last `parse_batch` (HIP events recorded on the stream each stage runs on; everything has
        been read back by now, so the events have completed): detector + hand-off graph, caption encode (crop -> DaViT -> BART encoder
        -> cross K/V of every micro-batch), the 20 decode steps.  The reference returns one wall-clock `latency` per request
        (ref:omnitool/omniparserserver/omniparserserver.py:42-44); the batch route adds this split."""
        ev, out = self._ev, {}
        for name, a, b in (("detect+handoff", "det0", "det1"), ("caption", "cap0", "capE"), ("caption_encode", "cap0", "cap1"),
                           ("caption_decode", "cap1", "cap2")):
            if a in ev and b in ev:
                try:
                    out[name] = round(float(ev[a].elapsed_time(ev[b])), 3)
                except RuntimeError:       # an event of an earlier batch that never completed on this path
                    pass
        self._ev = {}
        return out

    @staticmethod
    def _fill_captions(elems_all, caps):
        ids_out = []
        for el, cl in zip(elems_all, caps):
            q = list(cl)
            for e in el:
                if e["content"] is None and q:
                    e["content"] = q.pop(0)[0]
            ids_out.append([r for _, r in cl])
        return ids_out

    def _parse_batch_locked(self, frames, ocr, return_ids, iw, ih, pad_to=None):
        tiled = self.tile_large and (iw > 1952 or ih > 1112)
        handed = self.detect_glue(frames, ocr, pad_to) if (self.device_glue and not tiled) else None
        if handed is not None:
            dp, gs, ocr_els, counts = handed
            n_crops = [int(counts[f, 1]) for f in range(len(frames))]
            # crop rectangles were produced on the detector's stream, which the counts read-back above has drained: no event needed
            caps = self.caption(frames, n_crops, crops_dev=gs.crops)
            elems_all = self.assemble(dp, gs, ocr_els, counts, iw, ih, len(frames))
            ids_out = self._fill_captions(elems_all, caps)
            self.stats = {"crops": n_crops, "boxes": [int(v) for v in dp.out_count[: len(frames)].tolist()], "stage_ms": self.stage_ms(),
                          "split_overflow": getattr(self, "range_overflow_last", 0)}
            self.last_crops = [gs.crops[f, :n].tolist() for f, n in enumerate(n_crops)]
            return (elems_all, ids_out) if return_ids else elems_all
        if tiled:
            det_boxes = [self.detect_tiled(f)[0] for f in frames]      # >1080p: overlapping tiles + global NMS (our policy)
        else:
            det_boxes = self.detect(frames, pad_to)
        elems_all, crops_all = [], []
        for fi, xy in enumerate(det_boxes):
            texts, boxes = ocr[fi] if ocr is not None else ([], [])
            el, cr = self.glue(xy, iw, ih, boxes, texts)
            elems_all.append(el); crops_all.append(cr)
        caps = self.caption(frames, crops_all)
        ids_out = self._fill_captions(elems_all, caps)
        self.stats = {"crops": [len(c) for c in crops_all], "boxes": [len(b) for b in det_boxes], "split_overflow": getattr(self, "range_overflow_last", 0)}
        self.last_crops = crops_all            # integer crop boxes per frame, in caption order (parity tests read them)
        return (elems_all, ids_out) if return_ids else elems_all
