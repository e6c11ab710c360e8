"""Service layer: /parse/ wire format of ref:omnitool/omniparserserver/omniparserserver.py,
the /parse_batch/ extension, per-request OCR isolation under concurrency, and the client-side `reformat_messages`
(ref:omnitool/gradio/agent/llm_utils/omniparserclient.py:35-43).  Device adapters are stubs: this is host logic."""
import base64
import io
import threading
import types
import warnings
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch
from PIL import Image

from omniparser_amd import client as C
from omniparser_amd import server as S
from omniparser_amd.pipeline import ScreenParser
from omniparser_amd.synth import synthetic_screenshot
from omniparser_amd.util import omniparser as F
from omniparser_amd.util import utils as U


def _boxes_for(arr: np.ndarray) -> torch.Tensor:
    """Deterministic 'detections' derived from the image content and size."""
    h, w = arr.shape[:2]
    rng = np.random.default_rng(int(arr[::97, ::89].astype(np.int64).sum()) % (2 ** 31))
    n = 6 + int(rng.integers(0, 6))
    xy = rng.uniform(0.05, 0.8, (n, 2)) * [w, h]
    wh = rng.uniform(12, 60, (n, 2))
    return torch.tensor(np.concatenate([xy, xy + wh], 1), dtype=torch.float32)


class _Det:
    device = torch.device("cpu")
    _lock = threading.Lock()
    def predict(self, source, conf, iou, imgsz=None):
        b = _boxes_for(np.asarray(source))
        return [types.SimpleNamespace(boxes=types.SimpleNamespace(xyxy=b, conf=torch.ones(len(b))))]


class _Cap:
    config = types.SimpleNamespace(name_or_path="florence-fake", model_type="florence2")
    device = torch.device("cpu")
    _lock = threading.RLock()
    def caption_crops(self, image, boxes, max_new_tokens=20, batch_size=128):
        return torch.arange(len(boxes)).view(-1, 1)


class _Proc:
    def batch_decode(self, ids, skip_special_tokens=True): return [f" cap{int(i)} " for i in ids.view(-1)]


class _Screen(ScreenParser):
    """Real glue + real batch bookkeeping; device stages replaced by the same stubs the single-image path uses."""
    calls = 0
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.device_glue = False            # the stub detector has no plan to append the hand-off kernel to: host twin
    def detect(self, frames, pad_to=None):
        type(self).calls += 1
        return [_boxes_for(f.numpy()) for f in frames]
    def caption(self, frames, crops_per_frame, max_new_tokens=20):
        return [[(f"cap{k}", torch.tensor([k])) for k in range(len(c))] for c in crops_per_frame]


@pytest.fixture()
def service(monkeypatch):
    monkeypatch.setattr(U, "get_yolo_model", lambda model_path, device: _Det())
    monkeypatch.setattr(U, "get_caption_model_processor", lambda model_name, model_name_or_path, device: {"model": _Cap(), "processor": _Proc()})
    parser = F.Omniparser({"som_model_path": "x", "caption_model_name": "florence2", "caption_model_path": "y", "BOX_TRESHOLD": 0.05})
    _Screen.calls = 0
    screen = _Screen(parser.som_model, parser.caption_model_processor["model"], processor=_Proc(), box_threshold=0.05, iou_threshold=0.7)
    return S.ParseService(parser, screen_parser=screen, workers=4)


def _b64(seed, w, h):
    buf = io.BytesIO()
    Image.fromarray(synthetic_screenshot(seed, w, h)).save(buf, format="PNG")
    return base64.b64encode(buf.getvalue()).decode("ascii")


def _ocr(tag, w, h, n=3):
    return {"texts": [f"{tag}-{j}" for j in range(n)], "boxes": [[20 + 150 * j, h - 60, 120 + 150 * j, h - 30] for j in range(n)]}


def test_parse_route_wire_format(service):
    from fastapi.testclient import TestClient
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tc = TestClient(S.build_app({}, service=service))
    assert tc.get("/probe/").json() == {"message": "Omniparser API ready"}
    r = tc.post("/parse/", json={"base64_image": _b64(0, 640, 400), "ocr": _ocr("a", 640, 400)})
    assert r.status_code == 200
    body = r.json()
    assert set(body) == {"som_image_base64", "parsed_content_list", "latency"} and body["latency"] > 0
    assert Image.open(io.BytesIO(base64.b64decode(body["som_image_base64"]))).size == (640, 400)
    kinds = [e["type"] for e in body["parsed_content_list"]]
    assert kinds.count("text") == 3 and kinds.count("icon") >= 4 and kinds == sorted(kinds, key=lambda k: k != "text")
    assert tc.post("/parse/", json={"base64_image": _b64(0, 640, 400), "ocr": {"texts": ["a"], "boxes": []}}).status_code == 422
    assert tc.post("/parse/", json={}).status_code == 422


def test_batch_equals_single_and_keeps_request_order(service):
    sizes = [(640, 400), (800, 600), (640, 400), (640, 400), (480, 320), (800, 600)]
    items = [{"base64_image": _b64(s, w, h), "ocr": _ocr(f"r{s}", w, h) if s % 2 == 0 else None} for s, (w, h) in enumerate(sizes)]
    out = service.parse_many(items)
    assert len(out["results"]) == 6 and out["latency"] > 0
    assert _Screen.calls == 2                                  # {640x400 x3} and {800x600 x2} batched; 480x320 single
    for it, res in zip(items, out["results"]):
        single = service.parse_one(it["base64_image"], it["ocr"])
        assert res["parsed_content_list"] == single["parsed_content_list"]
        assert res["som_image_base64"] == single["som_image_base64"]     # same overlay, same PNG bytes
    assert S.ParseService.group_by_size([(1, 1)] * 19 + [(2, 2)], max_group=8) == [list(range(8)), list(range(8, 16)), [16, 17, 18], [19]]


def test_concurrent_requests_do_not_share_ocr(service):
    def one(k):
        w, h = 640, 400
        body = service.parse_one(_b64(k % 3, w, h), _ocr(f"req{k}", w, h))
        return k, [e["content"] for e in body["parsed_content_list"] if e["type"] == "text"]
    with ThreadPoolExecutor(8) as ex:
        for k, texts in ex.map(one, range(24)):
            assert texts == [f"req{k}-{j}" for j in range(3)]


def test_configured_ocr_provider_fills_batch_requests(service):
    service.parser.ocr_provider = lambda image: (["prov"], [[10, 10, 80, 30]])
    items = [{"base64_image": _b64(1, 640, 400)}, {"base64_image": _b64(2, 640, 400), "ocr": _ocr("own", 640, 400, 1)}]
    res = service.parse_many(items)["results"]
    assert [e["content"] for e in res[0]["parsed_content_list"] if e["type"] == "text"] == ["prov"]
    assert [e["content"] for e in res[1]["parsed_content_list"] if e["type"] == "text"] == ["own-0"]


def test_client_reformat_and_batch_url(service):
    elems = [{"type": "text", "content": "File"}, {"type": "icon", "content": "a gear"}, {"type": "other", "content": "?"},
             {"type": "icon", "content": "close"}]
    out = C.reformat_messages({"parsed_content_list": [dict(e) for e in elems]})
    assert out["screen_info"] == "ID: 0, Text: File\nID: 1, Icon: a gear\nID: 3, Icon: close\n"
    assert [e["idx"] for e in out["parsed_content_list"]] == [0, 1, 2, 3]
    seen = []
    def post(url, json):
        seen.append(url)
        body = service.parse_one(**json) if "base64_image" in json else service.parse_many(json["images"])
        return types.SimpleNamespace(status_code=200, json=lambda: body)
    cl = C.OmniParserClient("http://h:8000/parse/", post=post)
    img = Image.fromarray(synthetic_screenshot(4, 640, 400))
    r = cl(img, ocr=_ocr("c", 640, 400))
    assert (r["width"], r["height"]) == (640, 400) and r["screen_info"].startswith("ID: 0, Text: c-0\n")
    assert base64.b64decode(r["original_screenshot_base64"])[:4] == b"\x89PNG"
    rb = cl.parse_batch([img, img])
    assert seen == ["http://h:8000/parse/", "http://h:8000/parse_batch/"] and len(rb) == 2 and rb[0]["screen_info"] == rb[1]["screen_info"]
    assert [e["content"] for e in rb[0]["parsed_content_list"]][:2] == ["cap0", "cap1"]
