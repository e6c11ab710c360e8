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


# ---------------------------------------------------------------------------------------------------------------------------------
# OMNI_OP_PNG_DEFLATE: the same file with a COMPRESSED zlib stream, built so that every 4096-byte unit of the filtered scanline stream
# can be encoded by one GPU thread without knowing its neighbours' output:
#   * scanline filter: row 0 type 0 (None), every other row type 2 (Up: byte - byte above, mod 256) — flat regions become zeros;
#   * the filtered stream (filter bytes included) is cut into units of UNIT bytes; each unit is ONE deflate block with the FIXED
#     Huffman code (RFC 1951 3.2.6) holding literals and matches of distance 1 (byte runs) or 3 (pixel runs) found greedily
#     (longer of the two, length >= 3; matches may reach back across the unit boundary — the window is the decoder's), then an
#     EMPTY STORED block, which pads to a byte boundary (the "sync flush" marker 00 00 FF FF) so units concatenate bytewise;
#     a unit whose fixed-Huffman form is not smaller than UNIT + 5 bytes is emitted as one stored block instead;
#   * the last unit's trailing empty stored block carries BFINAL.
UNIT = 4096
_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]


def filtered_stream(frame: np.ndarray) -> np.ndarray:
    H, W, _ = frame.shape
    rows = frame.reshape(H, W * 3)
    out = np.zeros((H, W * 3 + 1), dtype=np.uint8)
    out[0, 1:] = rows[0]
    if H > 1:
        out[1:, 0] = 2
        out[1:, 1:] = rows[1:] - rows[:-1]          # uint8 arithmetic wraps mod 256
    return out.reshape(-1)


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):                    # LSB first
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def put_code(self, code, nbits):                # Huffman codes go in MSB first
        rev = 0
        for _ in range(nbits):
            rev = (rev << 1) | (code & 1)
            code >>= 1
        self.put(rev, nbits)

    def align(self):
        if self.n:
            self.out.append(self.acc & 255)
            self.acc = 0
            self.n = 0


def _lit_len_code(bits, sym):
    if sym < 144:
        bits.put_code(0x30 + sym, 8)
    elif sym < 256:
        bits.put_code(0x190 + sym - 144, 9)
    elif sym < 280:
        bits.put_code(sym - 256, 7)
    else:
        bits.put_code(0xC0 + sym - 280, 8)


def _match(bits, length, dist):
    k = max(j for j in range(29) if _LEN_BASE[j] <= length)
    _lit_len_code(bits, 257 + k)
    if _LEN_EXTRA[k]:
        bits.put(length - _LEN_BASE[k], _LEN_EXTRA[k])
    bits.put_code(0 if dist == 1 else 2, 5)


def deflate_unit(f: np.ndarray, start: int, end: int, final: bool) -> bytes:
    """one unit [start, end) of the filtered stream `f` -> its bytes in the zlib stream."""
    b = _Bits()
    b.put(0, 1)
    b.put(1, 2)                                     # BFINAL 0, BTYPE 01
    i = start
    while i < end:
        best, bd = 0, 0
        for dist in (1, 3):
            if i >= dist:
                l = 0
                while i + l < end and l < 258 and f[i + l] == f[i + l - dist]:
                    l += 1
                if l > best:
                    best, bd = l, dist
        if best >= 3:
            _match(b, best, bd)
            i += best
        else:
            _lit_len_code(b, int(f[i]))
            i += 1
    _lit_len_code(b, 256)
    b.put(1 if final else 0, 1)
    b.put(0, 2)                                     # empty stored block
    b.align()
    data = bytes(b.out) + b"\x00\x00\xff\xff"
    n = end - start
    if len(data) >= n + 5:
        data = struct.pack("<BHH", 1 if final else 0, n, n ^ 0xFFFF) + f[start:end].tobytes()
    return data


def deflate_png(frame: np.ndarray) -> bytes:
    H, W, C = frame.shape
    assert C == 3 and frame.dtype == np.uint8
    f = filtered_stream(frame)
    n = f.size
    z = bytearray(b"\x78\x01")
    for s in range(0, n, UNIT):
        z += deflate_unit(f, s, min(s + UNIT, n), s + UNIT >= n)
    z += struct.pack(">I", zlib.adler32(f.tobytes()) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", bytes(z)) + _chunk(b"IEND", b"")
