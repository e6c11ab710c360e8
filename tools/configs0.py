"""BASELINE configs[0] measured on both sides of the boundary: the reference's own demo image (tests/golden/ref_imgs/demo_image.jpg,
3240x2160, copied from /root/reference/imgs by tests/golden/gen_reference_images.py) through

  * the MI355X path behind the reference facade — `Omniparser(config).parse(base64)` (ref:util/omniparser.py:16-32): decode, OCR glue
    (synthetic OCR fixture), detector, hand-off, 768x768 caption crops, greedy decode, text, annotation + PNG — warm, N calls;
  * the reference-equivalent CPU pipeline on this box's host cores (oracle/: TorchScript YOLOv9-E stand-in in fp32, PIL letterbox,
    restated batched_nms; the same host glue; oracle crop pre-processing; transformers Florence-2 fp32 `generate` on EVERY crop at
    768x768 in the reference's 128-crop batches), ONE full pass — a measured run of configs[0], not a composition of samples.

usage (GPU box): python tools/configs0.py [--gpu-calls 5] [--no-cpu] > gpurun_out/.../configs0.json
The oracle is imported here as the timed CPU baseline and nowhere in the product (same rule as bench.py::cpu_baseline)."""
import argparse
import base64
import json
import sys
import time
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu-calls", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--image", default=str(ROOT / "tests" / "golden" / "ref_imgs" / "demo_image.jpg"))
    args = ap.parse_args()
    import torch
    from PIL import Image
    from omniparser_amd.synth import synthetic_ocr
    from omniparser_amd.util import utils as U
    from omniparser_amd.util.omniparser import Omniparser
    from tools.make_weights import build_random_captioner, ensure_blob, ensure_caption_checkpoint
    raw = Path(args.image).read_bytes()
    b64 = base64.b64encode(raw).decode("ascii")
    img = Image.open(args.image)
    iw, ih = img.size
    ocr = synthetic_ocr(7, iw, ih, 40)
    blob, cdir = ensure_blob(seed=0, nc=1, width=1.0), ensure_caption_checkpoint(0)
    out = {"image": Path(args.image).name, "size": [iw, ih], "workload": "BASELINE configs[0]: one image through Omniparser.parse (detect + caption at 768x768 crops, greedy, text, annotated PNG)"}
    # ---- MI355X
    cfg = {"som_model_path": str(blob), "caption_model_name": "florence2", "caption_model_path": str(cdir), "BOX_TRESHOLD": 0.05,
           "ocr_provider": lambda image: ocr, "caption_resolution": 768}
    op = Omniparser(cfg)
    t0 = time.perf_counter()
    png, elems = op.parse(b64)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(args.gpu_calls):
        png, elems = op.parse(b64)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / args.gpu_calls
    n_cap = sum(e["source"] == "box_yolo_content_yolo" for e in elems)
    out["mi355x"] = {"seconds_per_image": round(sec, 4), "images_per_s": round(1.0 / sec, 3), "calls": args.gpu_calls, "first_call_s": round(first, 2),
                     "elements": len(elems), "captioned_icons": n_cap, "png_base64_bytes": len(png),
                     "includes": "base64 + JPEG decode, OCR glue, H2D upload, detector, hand-off, captions, text, overlay + PNG encode + base64 (host Pillow)"}
    if args.no_cpu:
        print(json.dumps(out)); return
    # ---- CPU: the reference-equivalent pipeline, one full pass
    import gpu_checks as G
    from oracle import detector_ref as D
    cpu_model = torch.jit.load(str(blob), map_location="cpu").eval()
    ocap = G._OracleCaptioner(build_random_captioner(0), 768, use_cache=False)      # a timed CPU pass: never served from the oracle cache
    proc = U.FlorenceProcessor(cdir)

    class _Det:
        def predict(self, source, conf, iou, imgsz=None):
            b, s, c = D.predict(cpu_model, source, conf=conf, imgsz=imgsz or 640, iou=iou)
            return [types.SimpleNamespace(boxes=types.SimpleNamespace(xyxy=b, conf=s))]
    import io
    t0 = time.perf_counter()
    im2 = Image.open(io.BytesIO(base64.b64decode(b64)))
    (texts, boxes), _ = U.check_ocr_box(im2, ocr_result=ocr, display_img=False, output_bb_format="xyxy", easyocr_args={"text_threshold": 0.8}, use_paddleocr=False)
    t1 = time.perf_counter()
    enc_r, lab_r, el_r = U.get_som_labeled_img(im2, _Det(), BOX_TRESHOLD=0.05, output_coord_in_ratio=True, ocr_bbox=boxes, ocr_text=texts,
                                               use_local_semantics=True, iou_threshold=0.7, scale_img=False, batch_size=128,
                                               caption_model_processor={"model": ocap, "processor": proc})
    cpu_s = time.perf_counter() - t0
    out["cpu_reference_equivalent"] = {"seconds_per_image": round(cpu_s, 2), "images_per_s": round(1.0 / cpu_s, 5), "cores": torch.get_num_threads(),
                                       "kind": "port", "elements": len(el_r), "captioned_icons": len(ocap.boxes_seen),
                                       "sample": "ONE full pass of configs[0] (every crop captioned at 768x768 by transformers fp32 on the host cores)"}
    same = len(el_r) == len(elems) and all(a["type"] == b["type"] and a["source"] == b["source"] for a, b in zip(elems, el_r))
    caps_equal = sum(a["content"] == b["content"] for a, b in zip(elems, el_r) if a["source"] == "box_yolo_content_yolo") if same else None
    out["agreement"] = {"same_element_list_shape": bool(same), "captions_identical": caps_equal, "captioned": n_cap}
    out["speedup"] = round(cpu_s / sec, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
