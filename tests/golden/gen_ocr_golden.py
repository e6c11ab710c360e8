"""Golden fixture for the OCR front-end glue: executes the REFERENCE's own `check_ocr_box` (ref:util/utils.py:514-549) from
/root/reference under the dependency shims of gen_golden.py, with fake EasyOCR / PaddleOCR engines that return seeded
quads, and records (inputs, outputs).  The OCR MODELS themselves (CRAFT + CRNN inside EasyOCR, PP-OCR inside PaddleOCR) are
third-party packages that are absent here and out of this path's scope; what is pinned is everything the reference does
around them: RGBA handling, the PaddleOCR confidence filter (strict >, default 0.5), pass-through of `easyocr_args`, the
int() truncation of quad corners and the xywh / xyxy output formats (incl. the display_img=True branch, which always
returns xywh)."""
import json
import os
import sys
from pathlib import Path

import numpy as np
from PIL import Image

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))


def make_results(seed, n, w, h):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        x0, y0 = rng.uniform(0, w - 80), rng.uniform(0, h - 30)
        bw, bh = rng.uniform(8, 80), rng.uniform(6, 30)
        jitter = rng.uniform(-0.9, 0.9, size=(4, 2))
        quad = (np.array([[x0, y0], [x0 + bw, y0], [x0 + bw, y0 + bh], [x0, y0 + bh]]) + jitter).tolist()
        if i % 3 == 0:
            quad = [[int(a), int(b)] for a, b in quad]          # EasyOCR returns ints for axis-aligned boxes
        out.append((quad, f"text{i}", float(rng.uniform(0.05, 0.99))))
    return out


CASES = [
    dict(seed=0, n=12, size=(640, 480), mode="RGB", display_img=False, fmt="xywh", paddle=False, args=None),
    dict(seed=1, n=9, size=(1919, 1079), mode="RGBA", display_img=False, fmt="xyxy", paddle=False, args={"paragraph": False, "text_threshold": 0.9}),
    dict(seed=2, n=15, size=(800, 600), mode="RGB", display_img=False, fmt="xyxy", paddle=True, args=None),
    dict(seed=3, n=15, size=(800, 600), mode="RGB", display_img=False, fmt="xywh", paddle=True, args={"text_threshold": 0.8}),
    dict(seed=4, n=6, size=(320, 200), mode="RGB", display_img=True, fmt="xyxy", paddle=False, args=None),
    dict(seed=5, n=0, size=(320, 200), mode="RGB", display_img=False, fmt="xyxy", paddle=False, args=None),
]

if __name__ == "__main__":
    import gen_golden as G
    G.install_shims()
    import util.utils as U          # the reference module, from /root/reference
    records = []
    for c in CASES:
        res = make_results(c["seed"], c["n"], *c["size"])
        seen = {}

        class Reader:
            def readtext(self, image, **kw):
                seen["easyocr_kwargs"] = kw
                seen["shape"] = list(image.shape)
                return [(q, t, s) for q, t, s in res]

        class Paddle:
            def ocr(self, image, cls=False):
                seen["shape"] = list(image.shape)
                return [[[q, (t, s)] for q, t, s in res]]
        U.reader, U.paddle_ocr = Reader(), Paddle()
        img = Image.fromarray(np.zeros((c["size"][1], c["size"][0], 4 if c["mode"] == "RGBA" else 3), dtype=np.uint8), c["mode"])
        (text, bb), gf = U.check_ocr_box(img, display_img=c["display_img"], output_bb_format=c["fmt"], goal_filtering="gf",
                                         easyocr_args=c["args"], use_paddleocr=c["paddle"])
        records.append({"case": c, "results": res, "text": list(text), "bb": [list(map(int, b)) for b in bb], "goal_filtering": gf,
                        "engine_saw": seen})
    (HERE / "reference_ocr_glue.json").write_text(json.dumps(records, indent=0))
    print("wrote", len(records), "cases")
