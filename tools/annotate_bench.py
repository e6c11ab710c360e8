"""Time the tail of get_som_labeled_img (ref:util/utils.py:478-488: annotate + PNG + base64) on one 1920x1080 screenshot with a
realistic element count: the host path (numpy raster + Pillow PNG, `OMNI_OVERLAY=host`) against the device path (OMNI_OP_OVERLAY +
OMNI_OP_PNG_PACK, `OMNI_OVERLAY=device`), and check the device output on the way (same raster bytes; the PNG decodes to the frame).
Prints ONE JSON line.  bench.py runs it in a child process (`extra.annotate_tail`): `python tools/annotate_bench.py [--iters 5]`."""
import argparse
import base64
import io
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--elements", type=int, default=90)
    a = ap.parse_args()
    import numpy as np
    import torch
    from PIL import Image
    from omniparser_amd.synth import synthetic_screenshot
    from omniparser_amd.util import overlay as OV
    from omniparser_amd.util import utils as U
    dev = torch.device("cuda", 0)
    W, H, K = 1920, 1080, a.elements
    img = synthetic_screenshot(1, W, H)
    rng = np.random.default_rng(3)
    x1 = rng.integers(0, W - 200, K); y1 = rng.integers(0, H - 120, K)
    xyxy = np.stack([x1, y1, x1 + rng.integers(16, 200, K), y1 + rng.integers(16, 120, K)], 1).astype(np.float32)
    ratio = torch.from_numpy(xyxy / np.array([W, H, W, H], dtype=np.float32))
    boxes = U._box_convert_xyxy_to_cxcywh(ratio)
    phrases = list(range(K))

    def host():
        frame, coords = U.annotate(image_source=img, boxes=boxes, logits=None, phrases=phrases, text_scale=0.4, text_padding=5)
        return U.encode_png_b64(frame), frame

    def device():
        return U.annotate_encode_device(img, boxes, phrases, dev, text_scale=0.4, text_padding=5)[0]

    out = {"workload": f"annotate + PNG + base64 of one {W}x{H} screenshot, {K} elements (ref:util/utils.py:478-488)"}
    b64_h, frame_h = host()
    b64_d = device()
    torch.cuda.synchronize()
    frame_d = np.asarray(Image.open(io.BytesIO(base64.b64decode(b64_d))).convert("RGB"))
    out["device_png_decodes_to_host_raster"] = bool(np.array_equal(frame_d, frame_h))
    out["base64_bytes"] = {"host_pillow_png": len(b64_h), "device_deflate_png": len(b64_d), "stored_png": 4 * ((H * (3 * W + 1) + 5 * 95 + 63 + 2) // 3)}
    for name, fn in (("host_ms", host), ("device_ms", device)):
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        out[name] = round(1000 * (time.perf_counter() - t0) / a.iters, 2)
    # device kernels alone (frame resident, primitives built): HIP events
    cmds = OV.BoxAnnotator(text_scale=0.4, text_padding=5).plan(xyxy, [str(i) for i in range(K)], (W, H))
    fr = torch.from_numpy(img.copy()).to(dev)
    OV.render_device(fr, cmds); U.png_pack_device(fr); U.png_deflate_device(fr)
    torch.cuda.synchronize()
    U.png_deflate_device(fr, lz=False)
    torch.cuda.synchronize()
    e0, e1, e2, e3, e4 = (torch.cuda.Event(enable_timing=True) for _ in range(5))
    e0.record(); OV.render_device(fr, cmds); e1.record(); _, _, m_lz = U.png_deflate_device(fr); e2.record(); U.png_pack_device(fr); e3.record()
    _, _, m_fx = U.png_deflate_device(fr, lz=False); e4.record()
    torch.cuda.synchronize()
    out["device_kernels_ms"] = {"overlay (incl. table upload)": round(e0.elapsed_time(e1), 3),
                                "png_deflate LZ + dynamic Huffman + base64 (incl. scratch allocation)": round(e1.elapsed_time(e2), 3),
                                "png_pack (stored) + base64": round(e2.elapsed_time(e3), 3),
                                "png_deflate fixed Huffman + base64 (rounds 3-5)": round(e3.elapsed_time(e4), 3)}
    buf = io.BytesIO(); Image.fromarray(frame_h).save(buf, format="PNG")
    out["png_file_bytes"] = {"pillow_level6": len(buf.getvalue()), "device_lz_dynamic": int(m_lz[1].item()), "device_fixed_huffman": int(m_fx[1].item())}
    out["png_file_bytes"]["lz_over_pillow"] = round(out["png_file_bytes"]["device_lz_dynamic"] / out["png_file_bytes"]["pillow_level6"], 3)
    out["algorithmic_bytes"] = {"overlay": 2 * W * H * 3, "png_pack (stored) + base64": int(W * H * 3 * (1 + 1 + 1 + 1 + 4 / 3 + 4 / 3))}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
