"""FastAPI wrapper with the wire format of ref:omnitool/omniparserserver/omniparserserver.py
(POST /parse/ {base64_image} -> {som_image_base64, parsed_content_list, latency}; GET /probe/), so
OmniTool's `OmniParserClient` (ref:omnitool/gradio/agent/llm_utils/omniparserclient.py:14-33) talks to the
MI355X path unchanged.  The handler is a plain `def` (FastAPI runs it in its thread pool), unlike the
reference's `async def` that blocks the event loop.  OCR is out of scope: pass `--ocr-json` with
{"texts": [...], "boxes": [[x0,y0,x1,y1], ...]} per request field `ocr`, or run without text boxes.
"""
import argparse
import time


def build_app(config):
    from fastapi import FastAPI
    from pydantic import BaseModel
    from .util.omniparser import Omniparser

    app = FastAPI()
    parser = Omniparser(config)

    class ParseRequest(BaseModel):
        base64_image: str
        ocr: dict | None = None

    @app.post("/parse/")
    def parse(req: ParseRequest):
        start = time.time()
        if req.ocr:
            parser.ocr_provider = lambda image: (req.ocr.get("texts", []), req.ocr.get("boxes", []))
        som_image, parsed = parser.parse(req.base64_image)
        return {"som_image_base64": som_image, "parsed_content_list": parsed, "latency": time.time() - start}

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
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    a = ap.parse_args()
    import uvicorn
    uvicorn.run(build_app(vars(a)), host=a.host, port=a.port)


if __name__ == "__main__":
    main()
