"""`-m gpu`: the service layer on the real device path — the reference's `/parse/` route and the `/parse_batch/`
extension through FastAPI's test client (requests are served from worker threads, as in deployment), half-width detector, 64x64
caption crops.  The CPU suite covers the same routes with stub models (tests/test_service_cpu.py)."""
import base64
import io

import pytest

pytestmark = pytest.mark.gpu


def _b64(seed, w, h):
    from PIL import Image
    from omniparser_amd.synth import synthetic_screenshot
    buf = io.BytesIO()
    Image.fromarray(synthetic_screenshot(seed, w, h)).save(buf, format="PNG")
    return base64.b64encode(buf.getvalue()).decode("ascii")


def _ocr(seed, w, h):
    from omniparser_amd.synth import synthetic_ocr
    texts, boxes = synthetic_ocr(seed, w, h, 24)
    return {"texts": list(texts), "boxes": [list(map(float, b)) for b in boxes]}


def test_parse_and_parse_batch_routes_on_device():
    from fastapi.testclient import TestClient
    from omniparser_amd import server as S
    from tools.make_weights import ensure_blob, ensure_caption_checkpoint
    cfg = {"som_model_path": str(ensure_blob(seed=0, nc=1, width=0.5)), "caption_model_name": "florence2",
           "caption_model_path": str(ensure_caption_checkpoint(0)), "BOX_TRESHOLD": 0.05, "caption_resolution": 64, "device": "cuda"}
    client = TestClient(S.build_app(cfg))
    assert client.get("/probe/").json() == {"message": "Omniparser API ready"}
    sizes = [(2, 1280, 800), (3, 1280, 800), (1, 1920, 1080)]
    items = [{"base64_image": _b64(s, w, h), "ocr": _ocr(s, w, h)} for s, w, h in sizes]
    single = [client.post("/parse/", json=it) for it in items]
    assert all(r.status_code == 200 for r in single), [r.text[:300] for r in single]
    batch = client.post("/parse_batch/", json={"images": items})
    assert batch.status_code == 200, batch.text[:300]
    results = batch.json()["results"]
    assert len(results) == 3
    # the batch-2 group carries the HIP-event stage split of its parse_batch (device milliseconds of the stages' own streams)
    sm = results[0].get("stage_ms", {})
    assert sm.get("detect+handoff", 0) > 0 and sm.get("caption", 0) > 0, results[0].keys()
    for one, many, (s, w, h) in zip(single, results, sizes):
        a, b = one.json()["parsed_content_list"], many["parsed_content_list"]
        assert len(a) == len(b) > 10, (len(a), len(b))
        # the two 1280x800 frames share one batch-2 detector plan on the batch route and run alone on the single route: same kernels,
        # different tile schedules, so boxes may differ in the last bits and nothing else
        left, same = list(b), 0
        for ea in a:                          # order-free: two icons whose scores agree to the last bits may exchange ranks
            j = next((k for k, eb in enumerate(left) if max(abs(x - y) for x, y in zip(ea["bbox"], eb["bbox"])) <= 1e-5), None)
            assert j is not None, ea
            eb = left.pop(j)
            assert (ea["type"], ea["source"], ea["interactivity"]) == (eb["type"], eb["source"], eb["interactivity"])
            if ea["type"] == "text":
                assert ea["content"] == eb["content"]
            same += ea["content"] == eb["content"]
        assert same >= 0.9 * len(a), (same, len(a))
        assert any(e["source"] == "box_yolo_content_yolo" for e in a) and any(e["type"] == "text" for e in a)
        import PIL.Image as I
        img = I.open(io.BytesIO(base64.b64decode(many["som_image_base64"])))
        assert img.size == (w, h)
