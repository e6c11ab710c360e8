"""Overlay raster and PNG / base64 packing of csrc/overlay_png.hip, run from their device sources on the host emulation (tests/emu)
and compared with (a) the host raster `overlay.render`, itself compared with Pillow's ImageDraw here, (b) oracle/png_ref.py (layout
restatement on Python's zlib checksums), byte for byte, (c) Pillow's PNG reader.  The `-m gpu` twin: tests/test_gpu_f_overlay_png.py."""
import base64
import io

import numpy as np
import pytest
import torch
from PIL import Image, ImageDraw

from omniparser_amd.util import overlay as OV


def _scene(seed, W, H, K):
    rng = np.random.default_rng(seed)
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    x1 = rng.integers(-10, max(W - 20, 1), K); y1 = rng.integers(-10, max(H - 20, 1), K)
    bw = rng.integers(1, 90, K); bh = rng.integers(1, 60, K)
    bw[:3] = (1, 2, 5); bh[:3] = (1, 3, 2)                          # degenerate boxes: thinner than twice the stroke
    xyxy = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float64)
    cmds = OV.plan_overlay(xyxy, [str(i) for i in range(K)], (W, H), text_scale=0.4, text_padding=5)
    return frame, cmds


def test_host_raster_is_pillows_for_well_formed_primitives():
    """`render` (numpy execution of the primitive list) against ImageDraw itself: filled tags and label text everywhere, outlines
    wherever the rectangle is at least twice the stroke wide and high (Pillow's outline of smaller rectangles leaks outside them;
    ours fills them)."""
    frame, cmds = _scene(0, 320, 200, 40)
    ours = OV.render(frame.copy(), cmds)
    im = Image.fromarray(frame.copy())
    draw = ImageDraw.Draw(im)
    skipped = 0
    for c in cmds:
        if c[0] == OV.RECT:
            _, (x1, y1), (x2, y2), col, t = c
            if t == OV.FILLED:
                draw.rectangle([x1, y1, x2, y2], fill=col)
            else:
                o = t // 2
                if x2 - x1 + 2 * o + 1 < 2 * t or y2 - y1 + 2 * o + 1 < 2 * t:
                    skipped += 1
                    draw.rectangle([x1 - o, y1 - o, x2 + o, y2 + o], fill=col)
                else:
                    draw.rectangle([x1 - o, y1 - o, x2 + o, y2 + o], outline=col, width=t)
        else:
            _, text, (x, y), col, scale, _t = c
            draw.text((x, y), text, fill=col, font=OV._digit_font(scale), anchor="ls")
    assert skipped >= 2
    assert np.array_equal(ours, np.asarray(im))


@pytest.mark.parametrize("seed,W,H,K", [(1, 320, 200, 40), (2, 333, 97, 300), (3, 70, 30, 5)])
def test_device_raster_equals_host_raster(emu, seed, W, H, K):
    frame, cmds = _scene(seed, W, H, K)
    want = OV.render(frame.copy(), cmds)
    got = OV.render_device(torch.from_numpy(frame.copy()), cmds).numpy()
    assert np.array_equal(got, want)
    assert not np.array_equal(got, frame)


@pytest.mark.parametrize("H,W", [(37, 53), (300, 200), (1, 1), (130, 1000)])
def test_device_png_is_the_oracle_layout_and_decodes(emu, H, W):
    """file bytes == oracle/png_ref.py (so the device CRC-32 / Adler-32 equal zlib's), base64 == base64.b64encode, and Pillow reads
    the frame back.  (130 x 1000: six stored blocks, 96 CRC segments with a short first one.)"""
    from oracle import png_ref as PR
    from omniparser_amd.util.utils import png_pack_device
    frame = np.random.default_rng(H * W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    png, b64 = png_pack_device(torch.from_numpy(frame))
    data = png.numpy().tobytes()
    assert len(data) == PR.stored_png_size(H, W)
    assert data == PR.stored_png(frame)
    assert b64.numpy().tobytes() == base64.b64encode(data)
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), frame)


def _frames():
    from omniparser_amd.synth import synthetic_screenshot
    rng = np.random.default_rng(5)
    yield "noise", rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)                  # incompressible: stored fallback in every unit
    yield "flat", np.full((40, 300, 3), 77, dtype=np.uint8)                            # maximal runs (258-byte matches)
    yield "screenshot", synthetic_screenshot(1, 480, 270)
    yield "pixel runs", np.tile(rng.integers(0, 256, (1, 1, 3), dtype=np.uint8), (5, 3000, 1))   # distance-3 matches across units
    yield "one pixel", rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)
    f = rng.integers(0, 256, (64, 200, 3), dtype=np.uint8); f[::2] = f[1::2]            # every run length 3 .. 258 somewhere
    for k, l in enumerate(range(3, 259)):
        f[8 + k % 48, :, :].reshape(-1)[(k // 48) * 90:(k // 48) * 90 + l] = (k * 7) & 255
    yield "all match lengths", f


def _lz_frames():
    """what the LZ + dynamic-Huffman variant adds to `_frames`: repeated TEXTURE (hash matches at distances beyond 3, inside a unit and
    not across its start), more than one 32 KiB unit with a short last one, a unit where every byte value occurs (286-symbol code), a
    frequency profile skewed enough to need the 15-bit Kraft repair."""
    rng = np.random.default_rng(11)
    tile = rng.integers(0, 256, (8, 16, 3), dtype=np.uint8)
    yield "texture", np.tile(tile, (12, 40, 1))                                        # 96 x 640: two units, period 48 bytes
    g = np.zeros((48, 300, 3), dtype=np.uint8)
    g[..., 0] = np.arange(300)[None, :] % 251; g[..., 1] = (np.arange(48)[:, None] * 5) % 256; g[::7, ::3, 2] = 255
    yield "gradient", g
    sk = np.zeros((40, 280, 3), dtype=np.uint8)                                          # Fibonacci-like literal frequencies: deep Huffman tree
    flat = sk.reshape(-1)
    pos, a, b = 0, 1, 1
    for v in range(1, 24):
        flat[pos:pos + a * 3:3] = v; pos += a * 3; a, b = b, a + b
        if pos + b * 3 >= flat.size:
            break
    yield "skewed", sk


@pytest.mark.parametrize("name,frame", list(_frames()) + list(_lz_frames()), ids=[n for n, _ in list(_frames()) + list(_lz_frames())])
def test_device_lz_png_is_the_oracle_stream_and_decodes(emu, name, frame):
    """OMNI_OP_PNG_DEFLATE i5 = 1 (the device overlay's stream since round 6): file bytes == oracle/png_ref.py::deflate_png_lz — same
    tokens, same code lengths, same header, bit for bit — zlib inflates the stream to the filtered scanlines, Pillow reads the frame
    back, base64 matches; and the point of it: not larger than the fixed-Huffman variant, far smaller on texture."""
    import zlib
    from oracle import png_ref as PR
    from omniparser_amd.util.utils import png_deflate_device
    frame = np.ascontiguousarray(frame)
    png, b64, meta = png_deflate_device(torch.from_numpy(frame), lz=True)
    total, nb64 = int(meta[1]), int(meta[2])
    data = png[:total].numpy().tobytes()
    want = PR.deflate_png_lz(frame)
    assert total == len(want) and data == want, (total, len(want), next((k for k in range(min(total, len(want))) if data[k] != want[k]), None))
    assert zlib.decompress(data[41:41 + int(meta[0])]) == PR.filtered_stream(frame).tobytes()
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), frame)
    assert b64[:nb64].numpy().tobytes() == base64.b64encode(data)
    fixed = len(PR.deflate_png(frame))
    assert total <= fixed + 16, (total, fixed)
    if name == "texture":
        assert total < fixed // 4, (total, fixed)


def test_huffman_length_limit_repair_is_exercised_and_valid():
    """oracle/png_ref.py::huffman_lengths on frequencies that force a tree deeper than the limit: lengths stay within the limit and the
    Kraft sum is exactly 1 (a complete prefix code — what every inflater requires)."""
    from oracle import png_ref as PR
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    for freq, limit in ((fib, 15), (fib[:12], 7), ([5, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], 7), ([3, 0, 0, 9], 15)):
        lens = PR.huffman_lengths(freq, limit)
        used = [l for l in lens if l]
        assert max(used) <= limit and sum(2.0 ** -l for l in used) == 1.0, (freq, lens)
        assert all((l > 0) == (f > 0) for l, f in zip(lens, freq))


@pytest.mark.parametrize("name,frame", list(_frames()), ids=[n for n, _ in _frames()])
def test_device_deflate_png_is_the_oracle_stream_and_decodes(emu, name, frame):
    """OMNI_OP_PNG_DEFLATE, fixed-Huffman variant (lz=False): file bytes == oracle/png_ref.py::deflate_png (same filter, same greedy
    matcher, same fixed-Huffman bits, zlib's checksums), zlib inflates the stream to the filtered scanlines, Pillow reads the frame
    back, base64 matches."""
    import zlib
    from oracle import png_ref as PR
    from omniparser_amd.util.utils import png_deflate_device
    png, b64, meta = png_deflate_device(torch.from_numpy(np.ascontiguousarray(frame)), lz=False)
    total, nb64 = int(meta[1]), int(meta[2])
    data = png[:total].numpy().tobytes()
    want = PR.deflate_png(frame)
    assert total == len(want) and data == want
    z = data[41:41 + int(meta[0])]
    assert zlib.decompress(z) == PR.filtered_stream(frame).tobytes()
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), frame)
    assert b64[:nb64].numpy().tobytes() == base64.b64encode(data)
    if name in ("flat", "pixel runs"):
        assert total < frame.size // 20


def test_batch_route_renders_on_the_request_s_device_frame(emu, monkeypatch):
    """server.ParseService._render with OMNI_OVERLAY=device draws on the device copy of the screenshot that the batch route uploaded
    (no second upload) and returns a PNG that decodes to what the host raster + Pillow PNG of the same elements decodes to."""
    from omniparser_amd.server import ParseService
    rng = np.random.default_rng(9)
    rgb = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    elems = [{"bbox": [0.1, 0.2, 0.4, 0.5]}, {"bbox": [0.5, 0.1, 0.95, 0.9]}, {"bbox": [0.0, 0.0, 0.02, 0.03]}]
    svc = ParseService.__new__(ParseService)
    monkeypatch.setenv("OMNI_OVERLAY", "host")
    host = svc._render(rgb, elems)
    monkeypatch.setenv("OMNI_OVERLAY", "device")
    frame = torch.from_numpy(rgb.copy())
    dev = svc._render(rgb, elems, frame)
    a, b = (np.asarray(Image.open(io.BytesIO(base64.b64decode(x))).convert("RGB")) for x in (host, dev))
    assert a.shape == (120, 200, 3) and np.array_equal(a, b) and not np.array_equal(a, rgb)
    assert np.array_equal(frame.numpy(), a)                      # drawn in place on the request's frame
