"""HTTP service with the wire format of ref:omnitool/omniparserserver/omniparserserver.py:33-48
(POST /parse/ {base64_image} -> {som_image_base64, parsed_content_list, latency}; GET /probe/), so OmniTool's
`OmniParserClient` (ref:omnitool/gradio/agent/llm_utils/omniparserclient.py:14-33) talks to the MI355X path
unchanged — plus what a GPU service needs and the reference lacks:

* handlers are plain `def`s: FastAPI runs them on its thread pool instead of blocking the event loop the way
  the reference's `async def` does; the device plans serialise themselves (detector / captioner locks);
* `POST /parse_batch/ {images: [{base64_image, ocr?}, ...]}`: equally sized screenshots are parsed as ONE
  detector graph + packed caption micro-batches (`pipeline.ScreenParser.parse_batch`), PNG decode and the
  overlay + PNG encode of each image run on a host thread pool; response = {results: [<single-image schema>],
  latency};
* OCR is not part of the hot path: a request may carry `ocr: {"texts": [...], "boxes": [[x0,y0,x1,y1] px]}`
  per image (never stored on the shared parser — requests are concurrent).
"""
import argparse
import os
import time
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image

from .util.omniparser import Omniparser, decode_image, overlay_style


def _ocr_tuple(ocr: Optional[dict]):
    if not ocr:
        return None
    texts, boxes = list(ocr.get("texts", [])), [list(b) for b in ocr.get("boxes", [])]
    if len(texts) != len(boxes):
        raise ValueError(f"ocr: {len(texts)} texts but {len(boxes)} boxes")
    return texts, boxes


class ParseService:
    """Request-level logic, independent of the web framework (unit-tested with stub models)."""

    def __init__(self, parser: Omniparser, screen_parser=None, workers: int = 0):
        self.parser = parser
        self._screen = screen_parser           # pipeline.ScreenParser, built lazily from the facade's models
        self.pool = ThreadPoolExecutor(max_workers=workers or min(16, os.cpu_count() or 4), thread_name_prefix="omni-host")

    # ---- single image: the reference's route
    def parse_one(self, base64_image: str, ocr: Optional[dict] = None) -> dict:
        start = time.time()
        som_image, parsed = self.parser.parse(base64_image, ocr=_ocr_tuple(ocr))
        return {"som_image_base64": som_image, "parsed_content_list": parsed, "latency": time.time() - start}

    # ---- batch: device batching for equal-sized frames
    def screen_parser(self):
        if self._screen is None:
            from .pipeline import ScreenParser
            cmp = self.parser.caption_model_processor
            self._screen = ScreenParser(self.parser.som_model, cmp["model"], processor=cmp["processor"],
                                        box_threshold=self.parser.config["BOX_TRESHOLD"], iou_threshold=0.7, nms_iou=0.1,
                                        max_det=300, imgsz=640, batch_size=128)
        return self._screen

    @staticmethod
    def group_by_size(sizes: Sequence[Tuple[int, int]], max_group: int = 8) -> List[List[int]]:
        """Indices grouped by (w, h), arrival order kept inside a group, groups capped at `max_group` frames
        (one detector plan per (size, batch) — OMNI_MAX_DETECT_PLANS bounds how many stay resident)."""
        by = {}
        for i, s in enumerate(sizes):
            by.setdefault(tuple(s), []).append(i)
        groups = []
        for idx in by.values():
            groups += [idx[k:k + max_group] for k in range(0, len(idx), max_group)]
        return sorted(groups, key=lambda g: g[0])

    def _render(self, rgb: np.ndarray, elems: List[dict], frame_dev=None) -> str:
        import torch
        from .util import utils as U
        if os.environ.get("OMNI_SKIP_ANNOTATE", "0") == "1":
            return ""
        h, w = rgb.shape[:2]
        boxes = torch.tensor([e["bbox"] for e in elems], dtype=torch.float32).reshape(-1, 4)
        if frame_dev is not None and U.overlay_on_device(frame_dev.device):
            # raster + PNG + base64 on the device, on the copy of the screenshot this request already uploaded (csrc/overlay_png.hip)
            with torch.inference_mode():
                return U.annotate_encode_device(rgb, U._box_convert_xyxy_to_cxcywh(boxes), list(range(len(elems))), frame_dev.device,
                                                frame_dev=frame_dev, **overlay_style((w, h)))[0]
        frame, _ = U.annotate(rgb, U._box_convert_xyxy_to_cxcywh(boxes), None, list(range(len(elems))), **overlay_style((w, h)))
        return U.encode_png_b64(frame)

    def parse_many(self, items: Sequence[dict]) -> dict:
        import torch
        start = time.time()
        images = list(self.pool.map(lambda it: np.asarray(decode_image(it["base64_image"]).convert("RGB")), items))
        ocrs = [_ocr_tuple(it.get("ocr")) for it in items]
        if self.parser.ocr_provider is not None:            # configured provider fills what the request left out
            ocrs = [o if o is not None else self.parser.ocr_provider(Image.fromarray(im)) for o, im in zip(ocrs, images)]
        results: List[Optional[dict]] = [None] * len(items)
        for group in self.group_by_size([(im.shape[1], im.shape[0]) for im in images]):
            t0 = time.time()
            if len(group) == 1:
                i = group[0]
                png, elems = self.parser.parse_image(Image.fromarray(images[i]), ocrs[i])
                results[i] = {"som_image_base64": png, "parsed_content_list": elems, "latency": time.time() - t0}
                continue
            sp = self.screen_parser()
            frames = [torch.from_numpy(np.array(images[i], order="C")).to(sp.det.device) for i in group]
            elems = sp.parse_batch(frames, [ocrs[i] if ocrs[i] is not None else ([], []) for i in group])
            from .util import utils as U
            if U.overlay_on_device(sp.det.device):
                pngs = [self._render(images[i], el, fr) for i, el, fr in zip(group, elems, frames)]       # one stream: sequential
            else:
                pngs = list(self.pool.map(lambda a: self._render(*a), [(images[i], el) for i, el in zip(group, elems)]))
            dt = time.time() - t0
            stage_ms = dict(getattr(sp, "stats", {}).get("stage_ms", {}))      # HIP-event device time per stage of this group's batch
            for i, el, png in zip(group, elems, pngs):
                results[i] = {"som_image_base64": png, "parsed_content_list": el, "latency": dt, "stage_ms": stage_ms}
        return {"results": results, "latency": time.time() - start}


def build_app(config, service: Optional[ParseService] = None):
    from fastapi import FastAPI, HTTPException
    from pydantic import BaseModel

    app = FastAPI()
    svc = service or ParseService(Omniparser(config))

    class ParseRequest(BaseModel):
        base64_image: str
        ocr: Optional[dict] = None

    class BatchRequest(BaseModel):
        images: List[ParseRequest]

    @app.post("/parse/")
    def parse(req: ParseRequest):
        try:
            return svc.parse_one(req.base64_image, req.ocr)
        except ValueError as e:
            raise HTTPException(status_code=422, detail=str(e))

    @app.post("/parse_batch/")
    def parse_batch(req: BatchRequest):
        try:
            return svc.parse_many([{"base64_image": r.base64_image, "ocr": r.ocr} for r in req.images])
        except ValueError as e:
            raise HTTPException(status_code=422, detail=str(e))

    @app.get("/probe/")
    def probe():
        return {"message": "Omniparser API ready"}

    return app


def main():
    ap = argparse.ArgumentParser(description="Omniparser API (MI355X)")
    ap.add_argument("--som_model_path", default="weights/icon_detect_v3/model.pt")
    ap.add_argument("--caption_model_name", default="florence2")
    ap.add_argument("--caption_model_path", default="weights/icon_caption_florence")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--BOX_TRESHOLD", type=float, default=0.05)
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    a = ap.parse_args()
    import uvicorn
    uvicorn.run(build_app(vars(a)), host=a.host, port=a.port)


if __name__ == "__main__":
    main()
