"""Drop-in for ref:util/omniparser.py: same config keys, same `parse(image_base64)` contract."""
import base64
import io
from typing import Dict

import torch
from PIL import Image

from .utils import check_ocr_box, get_caption_model_processor, get_som_labeled_img, get_yolo_model


class Omniparser(object):
    def __init__(self, config: Dict):
        self.config = config
        device = "cuda" if torch.cuda.is_available() else "cpu"
        self.som_model = get_yolo_model(model_path=config.get("som_model_path"), device=device)
        self.caption_model_processor = get_caption_model_processor(
            model_name=config["caption_model_name"], model_name_or_path=config["caption_model_path"], device=device)
        self.ocr_provider = config.get("ocr_provider")   # callable(image) -> (texts, xyxy boxes); OCR itself is out of scope

    def parse(self, image_base64: str):
        image = Image.open(io.BytesIO(base64.b64decode(image_base64)))
        ratio = max(image.size) / 3200
        draw_bbox_config = {
            "text_scale": 0.8 * ratio,
            "text_thickness": max(int(2 * ratio), 1),
            "text_padding": max(int(3 * ratio), 1),
            "thickness": max(int(3 * ratio), 1),
        }
        ocr = self.ocr_provider(image) if self.ocr_provider else None
        (text, ocr_bbox), _ = check_ocr_box(image, display_img=False, output_bb_format="xyxy",
                                            easyocr_args={"text_threshold": 0.8}, use_paddleocr=False, ocr_result=ocr)
        labeled_img, label_coordinates, parsed_content_list = get_som_labeled_img(
            image, self.som_model, BOX_TRESHOLD=self.config["BOX_TRESHOLD"], output_coord_in_ratio=True, ocr_bbox=ocr_bbox,
            draw_bbox_config=draw_bbox_config, caption_model_processor=self.caption_model_processor, ocr_text=text,
            use_local_semantics=True, iou_threshold=0.7, scale_img=False, batch_size=128)
        return labeled_img, parsed_content_list
