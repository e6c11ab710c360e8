"""TEST ORACLE (imported by tests/ only): CPU restatement of the byte layout OMNI_OP_PNG_PACK (csrc/overlay_png.hip) emits for
ref:util/utils.py:485-488 (`PIL.Image.save(buf, format="PNG")` + `base64.b64encode`).  The reference's bytes depend on Pillow's
zlib settings and are not a contract — what the callers need is a valid PNG that decodes to the annotated frame — so the device
path fixes the simplest valid encoding (RFC 2083 / RFC 1950 / RFC 1951): 8-bit truecolour, filter type 0 on every scanline, ONE
IDAT chunk whose zlib stream (CMF/FLG 78 01) is a sequence of STORED deflate blocks of at most 65535 bytes, Adler-32, IEND.
Checksums come from Python's zlib, so this file also pins the device's own CRC-32 / Adler-32 arithmetic."""
import base64
import struct
import zlib

import numpy as np

STORED_MAX = 65535


def _chunk(kind: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def stored_png(frame: np.ndarray) -> bytes:
    """frame uint8 [H,W,3] -> PNG file bytes."""
    H, W, C = frame.shape
    assert C == 3 and frame.dtype == np.uint8
    raw = np.concatenate([np.zeros((H, 1), dtype=np.uint8), frame.reshape(H, W * 3)], 1).tobytes()     # filter byte 0 per scanline
    z = bytearray(b"\x78\x01")
    n = len(raw)
    for s in range(0, n, STORED_MAX):
        blk = raw[s:s + STORED_MAX]
        z += struct.pack("<BHH", 1 if s + STORED_MAX >= n else 0, len(blk), len(blk) ^ 0xFFFF) + blk
    z += struct.pack(">I", zlib.adler32(raw) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", bytes(z)) + _chunk(b"IEND", b"")


def stored_png_size(H: int, W: int) -> int:
    u = H * (3 * W + 1)
    return u + 5 * ((u + STORED_MAX - 1) // STORED_MAX) + 63


def stored_png_b64(frame: np.ndarray) -> str:
    return base64.b64encode(stored_png(frame)).decode("ascii")
