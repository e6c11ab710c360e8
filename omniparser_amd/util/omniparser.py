"""`Omniparser` facade of the MI355X path.

Mirrors the public contract of ref:util/omniparser.py:7-32 — the same config keys (`som_model_path`,
`caption_model_name`, `caption_model_path`, `BOX_TRESHOLD`), `parse(image_base64) -> (labeled_png_b64,
parsed_content_list)` — on top of the gfx950 detector / captioner adapters.  OCR is not part of the hot
path: an optional `ocr_provider` callable in the config supplies `(texts, xyxy_px_boxes)`;
without it the screenshot is parsed with icons only, exactly what the reference produces when its OCR
engine returns nothing.
"""
import base64
import io
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
from PIL import Image

from . import utils as U

# ref:util/omniparser.py:21-27 — overlay geometry grows linearly with the longer image side, 3200 px == 1.0
_OVERLAY_REF_SIDE = 3200
_OVERLAY_BASE = (("text_scale", 0.8, None), ("text_thickness", 2, 1), ("text_padding", 3, 1), ("thickness", 3, 1))
# fixed arguments the reference facade passes down (ref:util/omniparser.py:29-30)
_OCR_ARGS = dict(display_img=False, output_bb_format="xyxy", easyocr_args={"text_threshold": 0.8}, use_paddleocr=False)
_SOM_ARGS = dict(output_coord_in_ratio=True, use_local_semantics=True, iou_threshold=0.7, scale_img=False, batch_size=128)


def overlay_style(image_size: Tuple[int, int]) -> Dict[str, float]:
    """Annotation style for an image of `image_size` (w, h): integer fields are floored and clamped to >= 1."""
    ratio = max(image_size) / _OVERLAY_REF_SIDE
    return {name: (base * ratio if floor is None else max(int(base * ratio), floor)) for name, base, floor in _OVERLAY_BASE}


def decode_image(image_base64: str) -> Image.Image:
    return Image.open(io.BytesIO(base64.b64decode(image_base64)))


class Omniparser(object):
    def __init__(self, config: Dict):
        self.config = config
        device = "cuda" if torch.cuda.is_available() else "cpu"     # "cpu" makes the adapters raise: no CPU fallback
        self.som_model = U.get_yolo_model(model_path=config.get("som_model_path"), device=device)
        self.caption_model_processor = U.get_caption_model_processor(
            model_name=config["caption_model_name"], model_name_or_path=config["caption_model_path"], device=device)
        self.ocr_provider: Optional[Callable] = config.get("ocr_provider")
        # crop resolution of the captioner: 768 = the reference's CPU branch (bicubic to 768x768; the parity target and the
        # default here), 64 = its cuda branch (do_resize=False, ref:util/utils.py:120-121).  An explicit choice, not a device side effect.
        if config.get("caption_resolution") is not None:
            res = int(config["caption_resolution"])
            if res not in (64, 768):
                raise ValueError(f"caption_resolution must be 64 or 768, got {res}")
            self.caption_model_processor["model"].resolution = res

    def _ocr(self, image: Image.Image, ocr=None):
        """`ocr` = (texts, xyxy px boxes) handed over by the caller for THIS image; else the configured provider."""
        found = ocr if ocr is not None else (self.ocr_provider(image) if self.ocr_provider is not None else None)
        (texts, boxes), _ = U.check_ocr_box(image, ocr_result=found, **_OCR_ARGS)
        return texts, boxes

    def parse_image(self, image: Image.Image, ocr=None):
        texts, boxes = self._ocr(image, ocr)
        labeled, _coords, elements = U.get_som_labeled_img(
            image, self.som_model, BOX_TRESHOLD=self.config["BOX_TRESHOLD"], ocr_bbox=boxes, ocr_text=texts,
            draw_bbox_config=overlay_style(image.size), caption_model_processor=self.caption_model_processor, **_SOM_ARGS)
        return labeled, elements

    def parse(self, image_base64: str, ocr=None):
        return self.parse_image(decode_image(image_base64), ocr)

    def parse_many(self, images_base64: Sequence[str]):
        """Service helper: parse several screenshots with the same models (sequentially; the batched device path
        for equally sized frames is `omniparser_amd.pipeline.ScreenParser.parse_batch`)."""
        return [self.parse(b) for b in images_base64]
