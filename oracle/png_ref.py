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


# ---------------------------------------------------------------------------------------------------------------------------------
# OMNI_OP_PNG_DEFLATE, i5 = 1 (round 6; the default of the device overlay path): the same file with an LZ77 + DYNAMIC-Huffman stream,
# within ~1.1x (desktop screenshots) ... 1.3x (the noisy synthetic frames) of Pillow's zlib level 6 instead of 2.7x:
#   * the Up-filtered stream is cut into units of UNIT_LZ = 32768 bytes; ONE GPU lane encodes a unit, knowing nothing of its neighbours;
#   * tokens: at position i the candidates are distance 1 (byte runs), distance 3 (pixel runs) — both may reach back across the unit
#     start, the decoder's window holds those bytes — and the most recent earlier position INSIDE the unit whose 3 bytes hash to the same
#     13-bit bucket (one entry per bucket, every position inserted, matched spans included); the longest match wins (ties: the
#     smaller distance), minimum length 3, or 4 for a hash candidate further than 3 bytes away; greedy, no lazy evaluation;
#   * ONE dynamic-Huffman block per unit (RFC 1951 3.2.7): code lengths = Huffman over the unit's own literal/length and distance
#     frequencies (at least two symbols of each alphabet are given a code), limited to 15 bits by the usual Kraft repair (longest codes
#     first), canonical codes; the code-length sequence is run-length coded with symbols 16 / 17 / 18 and sent under a 7-bit-limited
#     code of its own; then an EMPTY STORED block (byte alignment: units concatenate bytewise; the last one carries BFINAL);
#   * a unit whose block is not smaller than its stored form (n + 5 bytes) is emitted stored.
# Every step below is written the way csrc/overlay_png.hip executes it (same scan orders, same tie-breaks): the device's bytes must
# equal these, bit for bit.
UNIT_LZ = 32768
HASH_BITS = 13
_DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289,
              16385, 24577]
_DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
_CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def _hash3(b0, b1, b2):
    return ((((b0 << 16) | (b1 << 8) | b2) * 0x9E3779B1) & 0xFFFFFFFF) >> (32 - HASH_BITS)


def lz_tokens(f, start, end):
    """[(length, distance)] with distance 0 = literal `length`."""
    table = [0] * (1 << HASH_BITS)              # position - start + 1 of the last 3-byte string per bucket, 0 = empty
    toks = []
    i = start

    def mlen(i, d):
        l = 0
        while i + l < end and l < 258 and f[i + l] == f[i + l - d]:
            l += 1
        return l

    def insert(q):
        if q + 2 < end:
            table[_hash3(int(f[q]), int(f[q + 1]), int(f[q + 2]))] = q - start + 1

    while i < end:
        best, bd = 0, 0
        if i >= 1:
            best, bd = mlen(i, 1), 1
        if i >= 3:
            l = mlen(i, 3)
            if l > best:
                best, bd = l, 3
        if i + 2 < end:
            c = table[_hash3(int(f[i]), int(f[i + 1]), int(f[i + 2]))]
            if c:
                d = i - (start + c - 1)
                l = mlen(i, d)
                if d > 3 and l < 4:
                    l = 0
                if l > best or (l == best and l > 0 and d < bd):
                    best, bd = l, d
        if best >= 3:
            toks.append((best, bd))
            for q in range(i, i + best):
                insert(q)
            i += best
        else:
            toks.append((int(f[i]), 0))
            insert(i)
            i += 1
    return toks


def _len_code(length):
    k = max(j for j in range(29) if _LEN_BASE[j] <= length)
    return k, _LEN_EXTRA[k], length - _LEN_BASE[k]


def _dist_code(dist):
    k = max(j for j in range(30) if _DIST_BASE[j] <= dist)
    return k, _DIST_EXTRA[k], dist - _DIST_BASE[k]


def huffman_lengths(freq, max_bits):
    """code length per symbol (0 = unused) — two-queue Huffman over the used symbols sorted by (frequency, symbol), then the Kraft
    repair that caps the lengths at `max_bits`, lengths re-dealt in that sorted order (rarest symbols get the longest codes)."""
    n = len(freq)
    used = sorted((freq[s], s) for s in range(n) if freq[s] > 0)
    m = len(used)
    lens = [0] * n
    if m == 0:
        return lens
    if m == 1:
        lens[used[0][1]] = 1
        return lens
    # nodes 0..m-1 = leaves in sorted order, m.. = internal nodes in creation order (non-decreasing weights)
    weight = [w for w, _ in used] + [0] * (m - 1)
    parent = [0] * (2 * m - 1)
    leaf, inode, nxt = 0, m, m
    for _ in range(m - 1):
        pick = []
        for _k in range(2):
            if leaf < m and (inode >= nxt or weight[leaf] <= weight[inode]):
                pick.append(leaf); leaf += 1
            else:
                pick.append(inode); inode += 1
        weight[nxt] = weight[pick[0]] + weight[pick[1]]
        parent[pick[0]] = parent[pick[1]] = nxt
        nxt += 1
    depth = [0] * (2 * m - 1)
    for k in range(2 * m - 3, -1, -1):
        depth[k] = depth[parent[k]] + 1
    count = [0] * (max_bits + 1)
    for k in range(m):
        count[min(depth[k], max_bits)] += 1
    total = sum(count[b] << (max_bits - b) for b in range(1, max_bits + 1))
    while total > (1 << max_bits):
        count[max_bits] -= 1
        for b in range(max_bits - 1, 0, -1):
            if count[b]:
                count[b] -= 1
                count[b + 1] += 2
                break
        total -= 1
    k = 0
    for b in range(max_bits, 0, -1):            # sorted order is ascending frequency: the longest codes go first
        for _ in range(count[b]):
            lens[used[k][1]] = b
            k += 1
    return lens


def canonical_codes(lens, max_bits):
    bl = [0] * (max_bits + 2)
    for l in lens:
        bl[l] += 1
    bl[0] = 0
    nxt, code = [0] * (max_bits + 2), 0
    for b in range(1, max_bits + 1):
        code = (code + bl[b - 1]) << 1
        nxt[b] = code
    codes = [0] * len(lens)
    for s, l in enumerate(lens):
        if l:
            codes[s] = nxt[l]
            nxt[l] += 1
    return codes


def _rle_code_lengths(seq):
    """[(symbol, extra bits, extra value)] over the code-length alphabet."""
    out = []
    i, n = 0, len(seq)
    while i < n:
        v = seq[i]
        j = i
        while j < n and seq[j] == v:
            j += 1
        run = j - i
        if v == 0:
            while run >= 11:
                r = min(run, 138); out.append((18, 7, r - 11)); run -= r
            if run >= 3:
                out.append((17, 3, run - 3)); run = 0
            out += [(0, 0, 0)] * run
        else:
            out.append((v, 0, 0)); run -= 1
            while run >= 3:
                r = min(run, 6); out.append((16, 2, r - 3)); run -= r
            out += [(v, 0, 0)] * run
        i = j
    return out


def deflate_unit_lz(f: np.ndarray, start: int, end: int, final: bool) -> bytes:
    toks = lz_tokens(f, start, end)
    lf, df = [0] * 286, [0] * 30
    for a, d in toks:
        if d == 0:
            lf[a] += 1
        else:
            lf[257 + _len_code(a)[0]] += 1
            df[_dist_code(d)[0]] += 1
    lf[256] += 1
    if sum(1 for v in lf if v) < 2:
        lf[0 if lf[0] == 0 else 1] += 1             # (only an empty unit: EOB alone) a second symbol so the code is complete
    for s in (0, 1):                                # at least two distance codes (RFC 1951 allows fewer; every decoder accepts two)
        if sum(1 for v in df if v) < 2 and df[s] == 0:
            df[s] = 1
    ll, dl = huffman_lengths(lf, 15), huffman_lengths(df, 15)
    lc, dc = canonical_codes(ll, 15), canonical_codes(dl, 15)
    hlit = max(s for s in range(286) if ll[s]) + 1
    hdist = max(s for s in range(30) if dl[s]) + 1
    hlit = max(hlit, 257)
    rle = _rle_code_lengths(ll[:hlit] + dl[:hdist])
    cf = [0] * 19
    for s, _, _ in rle:
        cf[s] += 1
    for s in (0, 18):
        if sum(1 for v in cf if v) < 2 and cf[s] == 0:
            cf[s] = 1
    cl = huffman_lengths(cf, 7)
    cc = canonical_codes(cl, 7)
    hclen = max(k for k in range(19) if cl[_CL_ORDER[k]]) + 1
    hclen = max(hclen, 4)
    b = _Bits()
    b.put(0, 1); b.put(2, 2)                        # BFINAL 0, BTYPE 10
    b.put(hlit - 257, 5); b.put(hdist - 1, 5); b.put(hclen - 4, 4)
    for k in range(hclen):
        b.put(cl[_CL_ORDER[k]], 3)
    for s, eb, ev in rle:
        b.put_code(cc[s], cl[s])
        if eb:
            b.put(ev, eb)
    for a, d in toks:
        if d == 0:
            b.put_code(lc[a], ll[a])
        else:
            k, eb, ev = _len_code(a)
            b.put_code(lc[257 + k], ll[257 + k])
            if eb:
                b.put(ev, eb)
            k, eb, ev = _dist_code(d)
            b.put_code(dc[k], dl[k])
            if eb:
                b.put(ev, eb)
    b.put_code(lc[256], ll[256])
    b.put(1 if final else 0, 1)
    b.put(0, 2)                                     # empty stored block
    b.align()
    data = bytes(b.out) + b"\x00\x00\xff\xff"
    n = end - start
    if len(data) >= n + 5:
        data = struct.pack("<BHH", 1 if final else 0, n, n ^ 0xFFFF) + f[start:end].tobytes()
    return data


def deflate_png_lz(frame: np.ndarray) -> bytes:
    H, W, C = frame.shape
    assert C == 3 and frame.dtype == np.uint8
    f = filtered_stream(frame)
    n = f.size
    z = bytearray(b"\x78\x01")
    for s in range(0, n, UNIT_LZ):
        z += deflate_unit_lz(f, s, min(s + UNIT_LZ, n), s + UNIT_LZ >= n)
    z += struct.pack(">I", zlib.adler32(f.tobytes()) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", bytes(z)) + _chunk(b"IEND", b"")
