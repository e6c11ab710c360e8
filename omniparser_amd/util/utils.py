"""Drop-in for the hot half of ref:util/utils.py on MI355X.

Same names, arguments and return values as the reference for the functions on the detect -> caption
path: `get_yolo_model`, `get_caption_model_processor`, `predict_yolo`,
`int_box_area`, `remove_overlap_new`, `get_parsed_content_icon`, `get_som_labeled_img`,
`check_ocr_box` (pass-through: OCR models are out of scope, SURVEY 2.1 #2b).  The two model stages run
as HIP plans (util/yolov9.py, florence.py); the glue between them reproduces the reference's list
semantics (App. E of SURVEY.md) and is pinned by golden fixtures generated from the reference's own
source under dependency shims (tests/golden/).
"""
import base64
import io
import os
import time
from pathlib import Path
from typing import List, Optional, Union

import numpy as np
import torch
from PIL import Image

from .. import _lib as L
from ..florence import CLIP_MEAN, CLIP_STD, PROMPT_IDS, Florence2Captioner
from .yolov9 import YOLOv9Detector


# ------------------------------------------------------------------------------------------ loaders
def get_yolo_model(model_path=None, device=None):
    """ref:util/utils.py:72-85 — local weights/icon_detect_v3/model.pt if present, YOLOv9-E adapter."""
    if model_path is None:
        local = Path(__file__).resolve().parents[2] / "weights/icon_detect_v3/model.pt"
        if local.is_file():
            model_path = local
    if model_path is None or "icon_detect_v3" in Path(model_path).parts:
        return YOLOv9Detector(model_path=model_path, device=device)
    raise NotImplementedError("only the YOLOv9-E icon_detect_v3 detector is implemented on MI355X "
                              "(the Ultralytics fallback of ref:util/utils.py:83-85 is out of scope)")


class _Batch(dict):
    """Minimal BatchFeature: mapping with `.to(device=, dtype=)` (ref:util/utils.py:121-123)."""

    def to(self, device=None, dtype=None, **kw):
        out = _Batch()
        for k, v in self.items():
            if isinstance(v, torch.Tensor):
                if dtype is not None and v.is_floating_point():
                    v = v.to(dtype)
                if device is not None:
                    v = v.to(device)
            out[k] = v
        return out

    __getattr__ = dict.get


class FlorenceProcessor:
    """Stand-in for AutoProcessor("microsoft/Florence-2-base") on the `<CAPTION>` path
    (hf:models/florence2/processing_florence2.py:84-152 + CLIP image processor).  Tokenizer files are
    not shipped with the reference; if `tokenizer.json` sits next to the checkpoint it is used for
    `batch_decode`, otherwise token ids are rendered as text."""

    def __init__(self, model_dir=None, image_token_id=51289, special_ids=(0, 1, 2, 3)):
        """special_ids: bos / pad / eos / unk of the checkpoint (Florence2Captioner passes what generation_config.json says) — used
        only WITHOUT a tokenizer.json; with one, the special ids are the tokens that file flags `special` (what
        `batch_decode(skip_special_tokens=True)` of the reference's AutoProcessor skips, ref:util/utils.py:128)."""
        self.image_token_id = image_token_id
        self.tok = None
        self.special = set(int(i) for i in special_ids) | {int(image_token_id)}
        if model_dir is not None and (Path(model_dir) / "tokenizer.json").exists():
            import json as _json
            from tokenizers import Tokenizer
            path = Path(model_dir) / "tokenizer.json"
            self.tok = Tokenizer.from_file(str(path))
            added = _json.loads(path.read_text()).get("added_tokens", [])
            self.special = {int(t["id"]) for t in added if t.get("special")} | {int(image_token_id)}
            self.vocab_size = self.tok.get_vocab_size(with_added_tokens=True)

    def __call__(self, images=None, text=None, return_tensors="pt", do_resize=True, **kw):
        imgs = images if isinstance(images, (list, tuple)) else [images]
        arrs = []
        for im in imgs:
            if do_resize:
                im = im.convert("RGB").resize((768, 768), Image.Resampling.BICUBIC)
            a = (np.asarray(im.convert("RGB")).astype(np.float64) * (1 / 255)).astype(np.float32)
            a = (a - np.asarray(CLIP_MEAN, dtype=np.float32)) / np.asarray(CLIP_STD, dtype=np.float32)
            arrs.append(torch.from_numpy(a.transpose(2, 0, 1).copy()))
        pix = torch.stack(arrs)
        n_img = (pix.shape[-1] // 32) ** 2 + 1
        ids = torch.tensor([[self.image_token_id] * n_img + PROMPT_IDS] * len(imgs))
        return _Batch(input_ids=ids, pixel_values=pix, attention_mask=torch.ones_like(ids))

    def batch_decode(self, ids, skip_special_tokens=True):
        special = self.special
        out = []
        for row in ids.tolist():
            toks = [t for t in row if not (skip_special_tokens and t in special)]
            if self.tok is not None:
                toks = [t for t in toks if 0 <= t < self.vocab_size]          # an id the tokenizer does not know (the image placeholder of another export)
                out.append(self.tok.decode(toks, skip_special_tokens=skip_special_tokens))
            else:
                out.append(" ".join(f"tok{t}" for t in toks))
        return out


def get_caption_model_processor(model_name, model_name_or_path="Salesforce/blip2-opt-2.7b", device=None):
    """ref:util/utils.py:48-69 for model_name == 'florence2' (BLIP-2 / phi3v are legacy, out of scope)."""
    if not device:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if model_name != "florence2":
        raise NotImplementedError(f"caption model '{model_name}': only 'florence2' is implemented on MI355X")
    model = Florence2Captioner(model_name_or_path, device)
    w = model.w
    processor = FlorenceProcessor(model_name_or_path, image_token_id=w.cfg.get("image_token_id", 51289),
                                  special_ids=(w.bos, w.pad, w.eos, w.cfg.get("text_config", {}).get("unk_token_id", 3)))
    return {"model": model, "processor": processor}


def get_xywh(input):
    """ref:util/utils.py:498-501 (quad corners -> truncated x, y, w, h)."""
    x, y, w, h = input[0][0], input[0][1], input[2][0] - input[0][0], input[2][1] - input[0][1]
    return int(x), int(y), int(w), int(h)


def get_xyxy(input):
    """ref:util/utils.py:503-506."""
    x, y, xp, yp = input[0][0], input[0][1], input[2][0], input[2][1]
    return int(x), int(y), int(xp), int(yp)


_OCR_ENGINES = {"easyocr": None, "paddleocr": None}      # process-wide engines, created on first use (the reference builds both at import)


def set_ocr_engine(easyocr_reader=None, paddle_ocr=None):
    """Install the OCR engine objects `check_ocr_box` calls: an EasyOCR-compatible `reader.readtext(image_np, **easyocr_args)` ->
    [(quad, text, conf)] and/or a PaddleOCR-compatible `ocr(image_np, cls=False)` -> [[[quad, (text, conf)], ...]]."""
    if easyocr_reader is not None:
        _OCR_ENGINES["easyocr"] = easyocr_reader
    if paddle_ocr is not None:
        _OCR_ENGINES["paddleocr"] = paddle_ocr


def _ocr_engine(kind):
    if _OCR_ENGINES[kind] is None:
        try:
            if kind == "easyocr":
                import easyocr
                _OCR_ENGINES[kind] = easyocr.Reader(["en"])                                  # ref:util/utils.py:21
            else:
                from paddleocr import PaddleOCR
                _OCR_ENGINES[kind] = PaddleOCR(lang="en", use_angle_cls=False, use_gpu=False, show_log=False, max_batch_size=1024,
                                               use_dilation=True, det_db_score_mode="slow", rec_batch_num=1024)   # ref:util/utils.py:22-30
        except ImportError:
            return None
    return _OCR_ENGINES[kind]


def check_ocr_box(image_source, display_img=True, output_bb_format="xywh", goal_filtering=None, easyocr_args=None,
                  use_paddleocr=False, ocr_result=None):
    """ref:util/utils.py:514-549 — the OCR front-end of every caller (`Omniparser.parse`, the Gradio demo).

    Everything the reference does AROUND its OCR engine is reproduced (fixture: tests/golden/reference_ocr_glue.json, recorded
    from the reference's own function): RGBA -> RGB, PaddleOCR's strict `> text_threshold` filter (default 0.5), `easyocr_args`
    handed to `readtext` untouched, int() truncation of the quad corners, xywh / xyxy output, and display_img=True always
    yielding xywh.  The engines themselves (EasyOCR = CRAFT + CRNN, PaddleOCR = PP-OCR) are third-party model packages that
    are not part of this path: they are used when importable or installed with `set_ocr_engine`; `ocr_result=(texts, xyxy
    boxes)` feeds precomputed OCR through the same formatting; with neither, the frame simply has no text boxes."""
    if isinstance(image_source, str):
        image_source = Image.open(image_source)
    if image_source.mode == "RGBA":
        image_source = image_source.convert("RGB")
    if ocr_result is not None:
        texts, boxes = ocr_result
        coord = [[[b[0], b[1]], [b[2], b[1]], [b[2], b[3]], [b[0], b[3]]] for b in boxes]
        text = list(texts)
    elif use_paddleocr:
        text_threshold = 0.5 if easyocr_args is None else easyocr_args["text_threshold"]
        engine = _ocr_engine("paddleocr")
        result = engine.ocr(np.array(image_source), cls=False)[0] if engine is not None else []
        coord = [item[0] for item in result if item[1][1] > text_threshold]
        text = [item[1][0] for item in result if item[1][1] > text_threshold]
    else:
        engine = _ocr_engine("easyocr")
        result = engine.readtext(np.array(image_source), **(easyocr_args or {})) if engine is not None else []
        coord = [item[0] for item in result]
        text = [item[1] for item in result]
    if display_img or output_bb_format == "xywh":        # the reference's display branch draws and keeps xywh
        bb = [get_xywh(item) for item in coord]
    elif output_bb_format == "xyxy":
        bb = [get_xyxy(item) for item in coord]
    else:
        raise UnboundLocalError(f"output_bb_format must be 'xywh' or 'xyxy', got {output_bb_format!r}")   # the reference leaves `bb` unbound
    return (text, bb), goal_filtering


# ------------------------------------------------------------------------------------------ glue (App. E)
def predict_yolo(model, image, box_threshold, imgsz, scale_img, iou_threshold=0.7):
    """ref:util/utils.py:388-409."""
    if scale_img:
        result = model.predict(source=image, conf=box_threshold, imgsz=imgsz, iou=iou_threshold)
    else:
        result = model.predict(source=image, conf=box_threshold, iou=iou_threshold)
    boxes = result[0].boxes.xyxy
    conf = result[0].boxes.conf
    phrases = [str(i) for i in range(len(boxes))]
    return boxes, conf, phrases


def int_box_area(box, w, h):
    """ref:util/utils.py:411-415."""
    x1, y1, x2, y2 = box
    ib = [int(x1 * w), int(y1 * h), int(x2 * w), int(y2 * h)]
    return (ib[2] - ib[0]) * (ib[3] - ib[1])


def _area(b):
    return (b[2] - b[0]) * (b[3] - b[1])


def _inter(a, b):
    return max(0, min(a[2], b[2]) - max(a[0], b[0])) * max(0, min(a[3], b[3]) - max(a[1], b[1]))


def _overlap(a, b):
    """the reference's IoU(): max(IoU with +1e-6 in the union, inter/areaA, inter/areaB)."""
    it = _inter(a, b)
    aa, ab = _area(a), _area(b)
    r1, r2 = (it / aa, it / ab) if (aa > 0 and ab > 0) else (0, 0)
    return max(it / (aa + ab - it + 1e-6), r1, r2)


def _inside(a, b):
    return _inter(a, b) / _area(a) > 0.80


def _remove_overlap_new_simple(boxes, iou_threshold, ocr_bbox=None):
    """Straightforward restatement of ref:util/utils.py:241-319 (kept as the cross-check of the vectorised
    version below; O(K^2 + K*M) Python)."""
    assert ocr_bbox is None or isinstance(ocr_bbox, list)
    out = list(ocr_bbox) if ocr_bbox else []
    bb = [e["bbox"] for e in boxes]
    for i, elem in enumerate(boxes):
        b1 = bb[i]
        a1 = _area(b1)
        if any(i != j and _overlap(b1, b2) > iou_threshold and a1 > _area(b2) for j, b2 in enumerate(bb)):
            continue
        if not ocr_bbox:
            out.append(b1)          # reference appends the bare box when there is no OCR list
            continue
        labels = ""
        swallowed = False
        for t in ocr_bbox:
            if _inside(t["bbox"], b1):
                labels += t["content"] + " "
                try:
                    out.remove(t)
                except ValueError:
                    pass
            elif _inside(b1, t["bbox"]):
                swallowed = True
                break
        if not swallowed:
            out.append({"type": "icon", "bbox": elem["bbox"], "interactivity": True,
                        "content": labels if labels else None,
                        "source": "box_yolo_content_ocr" if labels else "box_yolo_content_yolo"})
    return out


def _pair_inter(a, b):
    """f64 intersection areas [len(a), len(b)] with the reference's operation order."""
    iw = np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0])
    ih = np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1])
    return np.maximum(0, iw) * np.maximum(0, ih)


def remove_overlap_new(boxes, iou_threshold, ocr_bbox=None):
    """ref:util/utils.py:241-319 — same outputs and list-mutation quirks, vectorised (numpy f64 = the
    reference's Python-float arithmetic, same operation order) so the detect -> caption hand-off costs
    ~1 ms instead of ~100 ms of interpreter time at 300 boxes.

    An icon is dropped when some other icon overlaps it above the threshold and is smaller; OCR boxes
    lying inside a kept icon donate their text to it and are removed from the output (first dict-equal
    entry; a failed removal still donates the text); an icon lying inside an OCR box is dropped."""
    assert ocr_bbox is None or isinstance(ocr_bbox, list)
    out = list(ocr_bbox) if ocr_bbox else []
    n = len(boxes)
    if n == 0:
        return out
    bb = np.asarray([e["bbox"] for e in boxes], dtype=np.float64).reshape(n, 4)
    area = (bb[:, 2] - bb[:, 0]) * (bb[:, 3] - bb[:, 1])
    inter = _pair_inter(bb, bb)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / (area[:, None] + area[None, :] - inter + 1e-6)
        pos = (area[:, None] > 0) & (area[None, :] > 0)
        r1 = np.where(pos, inter / area[:, None], 0.0)
        r2 = np.where(pos, inter / area[None, :], 0.0)
    ov = np.maximum(np.maximum(iou, r1), r2)
    bad = (ov > iou_threshold) & (area[:, None] > area[None, :])
    np.fill_diagonal(bad, False)
    valid = ~bad.any(1)
    if not ocr_bbox:
        out.extend(boxes[i]["bbox"] for i in range(n) if valid[i])   # reference appends bare boxes without an OCR list
        return out
    m = len(ocr_bbox)
    ob = np.asarray([t["bbox"] for t in ocr_bbox], dtype=np.float64).reshape(m, 4)
    oarea = (ob[:, 2] - ob[:, 0]) * (ob[:, 3] - ob[:, 1])
    io = _pair_inter(bb, ob)                                     # [icons, ocr]
    ocr_in_icon = io / oarea[None, :] > 0.80                     # ZeroDivisionError in the reference <-> inf/nan here: never hit (int area > 0)
    icon_in_ocr = io / area[:, None] > 0.80
    # removal bookkeeping: `list.remove(x)` deletes the first remaining dict-equal entry
    key = [(tuple(t["bbox"]), t["content"], t["type"], t["interactivity"], t.get("source")) for t in ocr_bbox]
    alive = [True] * m
    members = {}
    for t_i, k in enumerate(key):
        members.setdefault(k, []).append(t_i)
    appended = []
    for i in np.nonzero(valid)[0]:
        stop = np.nonzero(~ocr_in_icon[i] & icon_in_ocr[i])[0]
        upto = stop[0] if len(stop) else m
        donors = np.nonzero(ocr_in_icon[i, :upto])[0]
        labels = ""
        for t_i in donors:
            labels += ocr_bbox[t_i]["content"] + " "
            for cand in members[key[t_i]]:
                if alive[cand]:
                    alive[cand] = False
                    break
        if len(stop):
            continue
        appended.append({"type": "icon", "bbox": boxes[i]["bbox"], "interactivity": True,
                         "content": labels if labels else None,
                         "source": "box_yolo_content_ocr" if labels else "box_yolo_content_yolo"})
    return [t for t_i, t in enumerate(ocr_bbox) if alive[t_i]] + appended


def crop_boxes_px(ratio_boxes, W, H):
    """Integer crop rectangles exactly as ref:util/utils.py:95-102 cuts them: `int(coord[k] * side)` there multiplies an element
    of an f32 TENSOR by a Python int — an f32 product, then truncation (NOT Python-float arithmetic: 100/1920 * 1920 is 100 in
    f32 and 99.99999 in f64).  Empty crops make cv2.resize raise and are skipped by the reference's bare `except`; numpy slicing
    clips the far edge at the image border.  Fixture: tests/golden/reference_glue.json::crop_shapes."""
    out = []
    fw, fh = np.float32(W), np.float32(H)
    for c in ratio_boxes:
        x0, x1 = int(np.float32(c[0]) * fw), int(np.float32(c[2]) * fw)
        y0, y1 = int(np.float32(c[1]) * fh), int(np.float32(c[3]) * fh)
        if x1 - x0 <= 0 or y1 - y0 <= 0 or x0 < 0 or y0 < 0:
            continue
        out.append([x0, y0, min(x1, W), min(y1, H)])
    return out


@torch.inference_mode()
def get_parsed_content_icon(filtered_boxes, starting_idx, image_source, caption_model_processor, prompt=None, batch_size=128):
    """ref:util/utils.py:88-132.  Crops are cut/resized/normalised on device and captioned by the HIP
    captioner; `image_source` may be the uint8 HWC numpy image (as in the reference) or a device tensor."""
    model, processor = caption_model_processor["model"], caption_model_processor["processor"]
    non_ocr = filtered_boxes[starting_idx:] if starting_idx else filtered_boxes
    H, W = image_source.shape[0], image_source.shape[1]
    boxes_px = crop_boxes_px(non_ocr.tolist() if isinstance(non_ocr, torch.Tensor) else non_ocr, W, H)
    if not boxes_px:
        return []
    img_dev = image_source if isinstance(image_source, torch.Tensor) else torch.from_numpy(np.array(image_source, order="C"))   # writable copy (PIL views are read-only)
    img_dev = img_dev.to(model.device)
    ids = model.caption_crops(img_dev, boxes_px, max_new_tokens=20, batch_size=batch_size)
    texts = processor.batch_decode(ids, skip_special_tokens=True)
    return [t.strip() for t in texts]


def _box_convert_xyxy_to_cxcywh(b: torch.Tensor) -> torch.Tensor:
    x1, y1, x2, y2 = b.unbind(-1)
    return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), -1)


def annotate(image_source: np.ndarray, boxes: torch.Tensor, logits, phrases, text_scale=0.4, text_padding=5,
             text_thickness=2, thickness=3):
    """ref:util/utils.py:336-364: boxes cxcywh ratios -> (annotated RGB frame, {str(phrase): xywh px}).  Labels are
    the running box indices (not `phrases`), exactly as the reference; layout = util/overlay.py (pinned to the
    reference's cv2 call sequence), raster = Pillow (cv2/supervision are absent)."""
    from .overlay import BoxAnnotator
    h, w, _ = image_source.shape
    b = boxes * torch.Tensor([w, h, w, h])
    cx, cy, bw, bh = b.unbind(-1)
    xyxy = torch.stack((cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh), -1).numpy()    # torchvision box_convert
    xywh = torch.stack((cx - 0.5 * bw, cy - 0.5 * bh, bw, bh), -1).numpy()
    labels = [f"{i}" for i in range(b.shape[0])]
    frame = np.array(image_source, order="C")
    BoxAnnotator(text_scale=text_scale, text_padding=text_padding, text_thickness=text_thickness, thickness=thickness).annotate(
        frame, xyxy, labels=labels, image_size=(w, h))
    label_coordinates = {f"{phrase}": v for phrase, v in zip(phrases, xywh)}
    return frame, label_coordinates


def encode_png_b64(frame: np.ndarray, compress_level: int = 6) -> str:
    """ref:util/utils.py:485-488: RGB frame -> PNG -> base64 ascii (compress_level: zlib effort, 6 = Pillow's default)."""
    buf = io.BytesIO()
    Image.fromarray(frame).save(buf, format="PNG", compress_level=compress_level)
    return base64.b64encode(buf.getvalue()).decode("ascii")


def overlay_on_device(device=None) -> bool:
    """Where the tail of `get_som_labeled_img` (ref:util/utils.py:478-488: annotate + PNG + base64) runs.  OMNI_OVERLAY = device | host
    decides; unset (round 6 on): on the device whenever the models live on a GPU — 18 ms instead of 51 ms per 1080p screenshot and a
    PNG within 1.16x of Pillow's bytes (tools/annotate_bench.py, profiles/r6_s4_annotate_bench.json) — on the host otherwise
    (rounds 2-5 defaulted to the host because the device PNG was 2.7x Pillow's size)."""
    mode = os.environ.get("OMNI_OVERLAY", "auto")
    if mode in ("device", "host"):
        return mode == "device"
    try:
        return device is not None and torch.device(device).type == "cuda"
    except (TypeError, RuntimeError):
        return False


def png_pack_device(frame: torch.Tensor, want_b64=True, stream=None):
    """OMNI_OP_PNG_PACK: uint8 [H,W,3] device tensor -> (PNG file bytes, base64 ASCII) as device tensors (stored-deflate PNG:
    include/omni_amd.h; layout restated in oracle/png_ref.py).  Five small launches, no host work."""
    H, W = frame.shape[:2]
    assert frame.dtype == torch.uint8 and frame.is_contiguous() and frame.shape[2] == 3
    u = H * (3 * W + 1)
    size = u + 5 * ((u + 65534) // 65535) + 63
    nseg = (size - 53 + 4095) // 4096
    png = torch.empty(size, dtype=torch.uint8, device=frame.device)
    part = torch.empty(2 * H + nseg, dtype=torch.int32, device=frame.device)
    b64 = torch.empty(4 * ((size + 2) // 3), dtype=torch.uint8, device=frame.device) if want_b64 else None
    L.launch(L.make_op(L.OP_PNG_PACK, L.F32, p=[frame.data_ptr(), png.data_ptr(), part.data_ptr(), b64.data_ptr() if want_b64 else None],
                       i={0: H, 1: W, 2: part.numel(), 3: size}), stream)
    return png, b64


def png_deflate_device(frame: torch.Tensor, want_b64=True, stream=None, lz=True):
    """OMNI_OP_PNG_DEFLATE: like png_pack_device with a compressed stream.  lz=True (default since round 6): Up filter + LZ77 +
    dynamic Huffman, one GPU lane per 32 KiB unit — within 1.1-1.3x of Pillow's zlib level 6 (oracle/png_ref.py::deflate_png_lz is
    the byte-exact restatement); lz=False: the fixed-Huffman run-length stream of rounds 3-5 (one thread per 4096-byte unit, 2.7x
    Pillow's bytes; oracle/png_ref.py::deflate_png).  The size is decided on the device:
    -> (PNG buffer, base64 buffer, meta) device tensors; meta[1] = file bytes, meta[2] = base64 bytes."""
    H, W = frame.shape[:2]
    assert frame.dtype == torch.uint8 and frame.is_contiguous() and frame.shape[2] == 3
    dev = frame.device
    u = H * (3 * W + 1)
    units = (u + 4095) // 4096
    units_lz = (u + 32767) // 32768
    cap = u + 5 * units + 63
    nseg = (cap - 53 + 4095) // 4096
    png = torch.empty(cap, dtype=torch.uint8, device=dev)
    filt = torch.empty(u, dtype=torch.uint8, device=dev)
    slots = torch.empty(max(units * 4640, units_lz * 33792 if lz else 0), dtype=torch.uint8, device=dev)
    meta = torch.zeros(4 + 2 * units, dtype=torch.int32, device=dev)
    part = torch.empty(2 * H + nseg, dtype=torch.int32, device=dev)
    b64 = torch.empty(4 * ((cap + 2) // 3), dtype=torch.uint8, device=dev) if want_b64 else None
    toks = torch.empty(units_lz * 32768, dtype=torch.int32, device=dev) if lz else None
    L.launch(L.make_op(L.OP_PNG_DEFLATE, L.F32,
                       p=[frame.data_ptr(), png.data_ptr(), filt.data_ptr(), slots.data_ptr(), meta.data_ptr(), part.data_ptr(),
                          b64.data_ptr() if want_b64 else None, toks.data_ptr() if lz else None],
                       i={0: H, 1: W, 2: cap, 3: meta.numel(), 4: part.numel(), 5: 1 if lz else 0}), stream)
    return png, b64, meta


def annotate_encode_device(image_np: np.ndarray, boxes: torch.Tensor, phrases, device, text_scale=0.4, text_padding=5, text_thickness=2,
                           thickness=3, frame_dev: Optional[torch.Tensor] = None, stored: bool = False):
    """`annotate` + `encode_png_b64` with the raster, the PNG packing and the base64 on the device (OMNI_OVERLAY=device): same
    layout (util/overlay.py::plan_overlay), same pixels as the host raster, stored-deflate PNG; the host uploads the frame and a
    few KB of primitives and reads back ASCII.  -> (base64 str, label_coordinates)."""
    from .overlay import BoxAnnotator, render_device
    h, w = (image_np.shape if image_np is not None else frame_dev.shape)[:2]
    b = boxes * torch.Tensor([w, h, w, h])
    cx, cy, bw, bh = b.unbind(-1)
    xyxy = torch.stack((cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh), -1).numpy()
    xywh = torch.stack((cx - 0.5 * bw, cy - 0.5 * bh, bw, bh), -1).numpy()
    ann = BoxAnnotator(text_scale=text_scale, text_padding=text_padding, text_thickness=text_thickness, thickness=thickness)
    cmds = ann.plan(xyxy, [f"{i}" for i in range(b.shape[0])], (w, h))
    # `frame_dev`: the caller's own uint8 [H,W,3] device copy of the screenshot (the batch route uploads one per request): drawn on
    # in place, no second upload
    frame = frame_dev if frame_dev is not None else torch.from_numpy(np.array(image_np, order="C")).to(device)
    render_device(frame, cmds)
    if stored:                                         # stored-deflate PNG (pixels + 0.02 %); default: the run-length deflate stream
        _, b64 = png_pack_device(frame)
        text = b64.cpu().numpy().tobytes()
    else:
        _, b64, meta = png_deflate_device(frame)
        n = int(meta[2].item())                        # the one synchronising read: the base64 length decided on the device
        text = b64[:n].cpu().numpy().tobytes()
    return text.decode("ascii"), {f"{phrase}": v for phrase, v in zip(phrases, xywh)}


def get_som_labeled_img(image_source: Union[str, Image.Image], model=None, BOX_TRESHOLD=0.01, output_coord_in_ratio=False,
                        ocr_bbox=None, text_scale=0.4, text_padding=5, draw_bbox_config=None, caption_model_processor=None,
                        ocr_text=[], use_local_semantics=True, iou_threshold=0.9, prompt=None, scale_img=False, imgsz=None,
                        batch_size=128):
    """ref:util/utils.py:417-496 — returns (base64 PNG, label_coordinates, filtered_boxes_elem)."""
    if isinstance(image_source, str):
        image_source = Image.open(image_source)
    image_source = image_source.convert("RGB")
    w, h = image_source.size
    if not imgsz:
        imgsz = (h, w)
    xyxy, logits, phrases = predict_yolo(model=model, image=image_source, box_threshold=BOX_TRESHOLD, imgsz=imgsz,
                                         scale_img=scale_img, iou_threshold=0.1)
    xyxy = xyxy.cpu() / torch.Tensor([w, h, w, h])       # f32 divide, as the reference (on its device)
    image_np = np.asarray(image_source)
    if ocr_bbox:
        ocr_bbox = (torch.tensor(ocr_bbox) / torch.Tensor([w, h, w, h])).tolist()
    else:
        ocr_bbox, ocr_text = [], []        # the reference raises TypeError here (zip(None, ...)); we tolerate it
    ocr_elems = [{"type": "text", "bbox": box, "interactivity": False, "content": txt, "source": "box_ocr_content_ocr"}
                 for box, txt in zip(ocr_bbox, ocr_text) if int_box_area(box, w, h) > 0]
    icon_elems = [{"type": "icon", "bbox": box, "interactivity": True, "content": None}
                  for box in xyxy.tolist() if int_box_area(box, w, h) > 0]
    filtered = remove_overlap_new(boxes=icon_elems, iou_threshold=iou_threshold, ocr_bbox=ocr_elems)
    if not ocr_elems:   # reference returns bare boxes in that (crashing) configuration; normalise to elements
        filtered = [e if isinstance(e, dict) else {"type": "icon", "bbox": e, "interactivity": True, "content": None,
                                                    "source": "box_yolo_content_yolo"} for e in filtered]
    elems = sorted(filtered, key=lambda x: x["content"] is None)
    starting_idx = next((i for i, box in enumerate(elems) if box["content"] is None), -1)
    filtered_boxes = torch.tensor([box["bbox"] for box in elems]).reshape(-1, 4)
    if use_local_semantics:
        parsed = get_parsed_content_icon(filtered_boxes, starting_idx, image_np, caption_model_processor, prompt=prompt,
                                         batch_size=batch_size)
        for box in elems:
            if box["content"] is None and parsed:
                box["content"] = parsed.pop(0)
    boxes_cxcywh = _box_convert_xyxy_to_cxcywh(filtered_boxes)
    phrases = [i for i in range(len(boxes_cxcywh))]
    if os.environ.get("OMNI_SKIP_ANNOTATE", "0") == "1":
        encoded = ""
        label_coordinates = {f"{i}": v for i, v in enumerate(
            torch.stack((filtered_boxes[:, 0] * w, filtered_boxes[:, 1] * h, (filtered_boxes[:, 2] - filtered_boxes[:, 0]) * w,
                         (filtered_boxes[:, 3] - filtered_boxes[:, 1]) * h), -1).numpy())} if len(filtered_boxes) else {}
    else:
        cfg = draw_bbox_config or {"text_scale": text_scale, "text_padding": text_padding}
        if overlay_on_device(getattr(model, "device", None)):        # raster + PNG + base64 on the MI355X (csrc/overlay_png.hip)
            encoded, label_coordinates = annotate_encode_device(image_np, boxes_cxcywh, phrases, model.device, **cfg)
        else:
            frame, label_coordinates = annotate(image_source=image_np, boxes=boxes_cxcywh, logits=logits, phrases=phrases, **cfg)
            encoded = encode_png_b64(frame)
    if output_coord_in_ratio:
        label_coordinates = {k: [v[0] / w, v[1] / h, v[2] / w, v[3] / h] for k, v in label_coordinates.items()}
    return encoded, label_coordinates, elems
