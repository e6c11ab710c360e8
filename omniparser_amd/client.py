"""Client side of the /parse/ wire format (ref:omnitool/gradio/agent/llm_utils/omniparserclient.py:9-43).

The reference's client grabs a screenshot from the OmniBox VM, posts it and re-shapes the answer for the agent
loop; screen capture belongs to OmniTool, so here the caller supplies the image.  The response post-processing
(`reformat_messages`: running `idx` per element + the `screen_info` text block the LLM prompt embeds) follows the
reference; `parse_batch` talks to this service's /parse_batch/ extension.
"""
import base64
import io
from pathlib import Path
from typing import Callable, Optional, Sequence, Union

from PIL import Image

ImageLike = Union[str, Path, bytes, Image.Image]


def encode_image(image: ImageLike) -> str:
    """PNG/JPEG file, raw encoded bytes or a PIL image -> base64 ascii of an encoded image file."""
    if isinstance(image, Image.Image):
        buf = io.BytesIO()
        image.save(buf, format="PNG")
        raw = buf.getvalue()
    elif isinstance(image, (bytes, bytearray)):
        raw = bytes(image)
    else:
        raw = Path(image).read_bytes()
    return base64.b64encode(raw).decode("ascii")


def screen_info(parsed_content_list: Sequence[dict]) -> str:
    """One line per element, `ID: <idx>, Text: <content>` / `ID: <idx>, Icon: <content>`; other types are skipped
    but still consume an index (ref:...omniparserclient.py:35-43)."""
    kinds = {"text": "Text", "icon": "Icon"}
    return "".join(f"ID: {i}, {kinds[e['type']]}: {e['content']}\n" for i, e in enumerate(parsed_content_list) if e["type"] in kinds)


def reformat_messages(response_json: dict) -> dict:
    for idx, element in enumerate(response_json["parsed_content_list"]):
        element["idx"] = idx
    response_json["screen_info"] = screen_info(response_json["parsed_content_list"])
    return response_json


class OmniParserClient:
    def __init__(self, url: str, post: Optional[Callable] = None):
        """`url` = the /parse/ endpoint; `post(url, json=...)` defaults to `requests.post` (injectable for tests)."""
        self.url = url
        if post is None:
            import requests
            post = requests.post
        self._post = post

    def _finish(self, response_json: dict, image_base64: str, size) -> dict:
        response_json["width"], response_json["height"] = size
        response_json["original_screenshot_base64"] = image_base64
        return reformat_messages(response_json)

    @staticmethod
    def _size(image: ImageLike, image_base64: str):
        if isinstance(image, Image.Image):
            return image.size
        return Image.open(io.BytesIO(base64.b64decode(image_base64))).size

    def __call__(self, image: ImageLike, ocr: Optional[dict] = None) -> dict:
        b64 = encode_image(image)
        payload = {"base64_image": b64}
        if ocr is not None:
            payload["ocr"] = ocr
        resp = self._post(self.url, json=payload)
        if getattr(resp, "status_code", 200) != 200:
            raise RuntimeError(f"{self.url} -> HTTP {resp.status_code}: {getattr(resp, 'text', '')[:200]}")
        return self._finish(resp.json(), b64, self._size(image, b64))

    def parse_batch(self, images: Sequence[ImageLike], ocr: Optional[Sequence[Optional[dict]]] = None) -> list:
        b64 = [encode_image(im) for im in images]
        items = [{"base64_image": b} if not (ocr and ocr[i]) else {"base64_image": b, "ocr": ocr[i]} for i, b in enumerate(b64)]
        url = self.url.rstrip("/")
        url = (url[: -len("/parse")] if url.endswith("/parse") else url) + "/parse_batch/"
        resp = self._post(url, json={"images": items})
        if getattr(resp, "status_code", 200) != 200:
            raise RuntimeError(f"{url} -> HTTP {resp.status_code}: {getattr(resp, 'text', '')[:200]}")
        return [self._finish(r, b, self._size(im, b)) for r, b, im in zip(resp.json()["results"], b64, images)]
