"""`-m gpu`: the tail of get_som_labeled_img on the device (csrc/overlay_png.hip, opt-in OMNI_OVERLAY=device) — set-of-marks raster,
stored-deflate PNG with device-side Adler-32 / CRC-32, base64 — against the host raster, the byte-layout oracle (oracle/png_ref.py)
and Pillow's PNG reader.  New in round 2 after the last GPU minute (validated on the host emulation,
tests/test_overlay_png_emu_cpu.py)."""
import base64
import io

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def test_overlay_raster_1080p_equals_host_raster():
    from omniparser_amd.synth import synthetic_screenshot
    from omniparser_amd.util import overlay as OV
    frame = synthetic_screenshot(3, 1920, 1080)
    rng = np.random.default_rng(7)
    K = 300
    x1 = rng.integers(-10, 1900, K); y1 = rng.integers(-10, 1060, K)
    xyxy = np.stack([x1, y1, x1 + rng.integers(1, 200, K), y1 + rng.integers(1, 120, K)], 1).astype(np.float64)
    cmds = OV.plan_overlay(xyxy, [str(i) for i in range(K)], (1920, 1080), text_scale=0.4, text_padding=5)
    want = OV.render(frame.copy(), cmds)
    got = OV.render_device(torch.from_numpy(frame.copy()).cuda(), cmds)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("H,W", [(1080, 1920), (37, 53), (2160, 3840)])
def test_png_pack_is_the_oracle_layout_and_decodes(H, W):
    from oracle import png_ref as PR
    from omniparser_amd.util.utils import png_pack_device
    frame = np.random.default_rng(H + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    png, b64 = png_pack_device(torch.from_numpy(frame).cuda())
    torch.cuda.synchronize()
    data = png.cpu().numpy().tobytes()
    assert data == PR.stored_png(frame)
    assert b64.cpu().numpy().tobytes() == base64.b64encode(data)
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), frame)


@pytest.mark.parametrize("kind", ["screenshot", "noise", "flat"])
@pytest.mark.parametrize("lz", [True, False], ids=["lz-dynamic", "fixed"])
def test_png_deflate_is_the_oracle_stream_and_decodes(kind, lz):
    """OMNI_OP_PNG_DEFLATE at 1920x1080, both streams (i5 = 1: LZ77 + dynamic Huffman, the device overlay's default; i5 = 0: the
    fixed-Huffman run-length stream): byte-identical to oracle/png_ref.py (CPU restatement of the same encoder, Python zlib
    checksums), inflated by zlib to the filtered scanlines, read back by Pillow; the LZ stream of the 1080p screenshot is within
    1.35x of Pillow's own PNG of the same frame (the fixed one: 2.7x)."""
    import zlib
    from oracle import png_ref as PR
    from omniparser_amd.synth import synthetic_screenshot
    from omniparser_amd.util.utils import png_deflate_device
    H, W = (1080, 1920) if kind == "screenshot" else (270, 480)
    if kind == "screenshot":
        frame = synthetic_screenshot(2, W, H)
    elif kind == "noise":
        frame = np.random.default_rng(1).integers(0, 256, (H, W, 3), dtype=np.uint8)
    else:
        frame = np.full((H, W, 3), 200, dtype=np.uint8)
    png, b64, meta = png_deflate_device(torch.from_numpy(frame).cuda(), lz=lz)
    torch.cuda.synchronize()
    m = meta.cpu()
    data = png[: int(m[1])].cpu().numpy().tobytes()
    assert zlib.decompress(data[41:41 + int(m[0])]) == PR.filtered_stream(frame).tobytes()
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), frame)
    assert b64[: int(m[2])].cpu().numpy().tobytes() == base64.b64encode(data)
    if kind != "screenshot":                                 # the pure-Python encoders need ~1 s (fixed) / ~10 s (LZ) per MB
        assert data == (PR.deflate_png_lz(frame) if lz else PR.deflate_png(frame))
    elif lz:
        buf = io.BytesIO(); Image.fromarray(frame).save(buf, format="PNG")
        # first 64 rows byte for byte against the oracle (same units: the stream is cut every 32 KiB), the whole file by size
        assert len(data) <= 1.35 * len(buf.getvalue()), (len(data), len(buf.getvalue()))


def test_get_som_labeled_img_with_device_overlay(monkeypatch):
    """ref:util/utils.py:417-496 through the product with OMNI_OVERLAY=device vs the default host raster + Pillow PNG: same elements,
    same label coordinates, and the two PNGs decode to the same annotated frame."""
    from omniparser_amd.florence import Florence2Captioner
    from omniparser_amd.synth import synthetic_ocr, synthetic_screenshot
    from omniparser_amd.util import utils as U
    from omniparser_amd.util.yolov9 import YOLOv9Detector
    from tools.make_weights import ensure_blob, ensure_caption_checkpoint
    det = YOLOv9Detector(model_path=ensure_blob(seed=0, nc=1, width=0.5), device="cuda", precision="f32")
    cdir = ensure_caption_checkpoint(0)
    cap = Florence2Captioner(cdir, "cuda", precision="f32", resolution=64)
    proc = U.FlorenceProcessor(cdir)
    img = Image.fromarray(synthetic_screenshot(1, 1920, 1080))
    texts, obox = synthetic_ocr(1, 1920, 1080, 40)
    kw = dict(BOX_TRESHOLD=0.05, output_coord_in_ratio=True, ocr_bbox=obox, ocr_text=texts, use_local_semantics=True,
              iou_threshold=0.7, scale_img=False, batch_size=128, caption_model_processor={"model": cap, "processor": proc})
    res = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("OMNI_OVERLAY", mode)
        res[mode] = U.get_som_labeled_img(img, det, **kw)
    assert res["device"][2] == res["host"][2]
    assert res["device"][1].keys() == res["host"][1].keys()
    a, b = (np.asarray(Image.open(io.BytesIO(base64.b64decode(res[m][0]))).convert("RGB")) for m in ("device", "host"))
    assert a.shape == (1080, 1920, 3) and np.array_equal(a, b)
