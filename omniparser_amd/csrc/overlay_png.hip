// Set-of-marks overlay and PNG / base64 packing on the device: the tail of get_som_labeled_img (ref:util/utils.py:478-491 —
// annotate(), PIL save to PNG, base64) without a host raster or a host deflate.  Byte work, HBM-bound: every kernel touches each
// byte of the 6.2 MB frame (1920x1080x3) once or twice.
//
//   OMNI_OP_OVERLAY   painter's algorithm per pixel over the primitive list of util/overlay.py::raster_primitives (filled
//                     rectangle, ring = outline of a given width inside a rectangle, 8-bit coverage mask blended like Pillow's
//                     draw_bitmap): one thread per pixel walks the primitives IN ORDER; a workgroup (64x4 pixels) first culls
//                     them against its tile, 256 at a time, into LDS flags.
//   OMNI_OP_PNG_PACK  frame -> complete PNG file image in device memory: signature, IHDR, one IDAT holding a zlib stream of
//                     STORED deflate blocks (filter byte 0 per scanline), IEND; Adler-32 and CRC-32 are computed on the device
//                     (per-segment partials, combined by one thread with the GF(2) shift operator x^(8 len) mod P), then the whole
//                     file is base64-encoded.  Five small launches on one stream; the host reads back ASCII.
// The byte layout is restated on the CPU in oracle/png_ref.py (test oracle); any PNG reader is the second check.
#include "omni_internal.h"

namespace {

// ------------------------------------------------------------------------------------ overlay
struct OvArgs {
  unsigned char* img; const int* prim; const unsigned char* masks;
  int H, W, n;
};
constexpr int PRIM_INTS = 8;     // {kind, x0, y0, x1, y1, r | g << 8 | b << 16, a, b}
enum { PRIM_FILL = 0, PRIM_RING = 1, PRIM_MASK = 2 };

// Pillow's BLEND8 / DIV255: round(in * (255 - m) / 255 + ink * m / 255) in its integer form
__device__ __forceinline__ unsigned blend8(unsigned m, unsigned in, unsigned ink) {
  const unsigned t = in * (255u - m) + ink * m + 128u;
  return (t + (t >> 8)) >> 8;
}

__global__ __launch_bounds__(256) void overlay_kernel(OvArgs a) {
  __shared__ int sp[256 * PRIM_INTS];
  __shared__ unsigned char hit[256];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int bx0 = blockIdx.x * 64, by0 = blockIdx.y * 4;
  const int x = bx0 + tx, y = by0 + ty;
  const bool live = x < a.W && y < a.H;
  unsigned r = 0, g = 0, b = 0;
  unsigned char* px = a.img + ((long long)y * a.W + x) * 3;
  if (live) { r = px[0]; g = px[1]; b = px[2]; }
  bool dirty = false;
  for (int base = 0; base < a.n; base += 256) {
    const int c = base + (int)threadIdx.x;
    unsigned char h = 0;
    if (c < a.n) {
      const int* p = a.prim + (long long)c * PRIM_INTS;
#pragma unroll
      for (int q = 0; q < PRIM_INTS; ++q) sp[threadIdx.x * PRIM_INTS + q] = p[q];
      // bounding box of the primitive (masks: x1 / y1 exclusive) against this tile
      const int kx1 = p[0] == PRIM_MASK ? p[3] - 1 : p[3], ky1 = p[0] == PRIM_MASK ? p[4] - 1 : p[4];
      h = (p[1] <= bx0 + 63 && kx1 >= bx0 && p[2] <= by0 + 3 && ky1 >= by0) ? 1 : 0;
    }
    hit[threadIdx.x] = h;
    __syncthreads();
    const int cnt = min(256, a.n - base);
    if (live) {
      for (int q = 0; q < cnt; ++q) {
        if (!hit[q]) continue;
        const int* p = sp + q * PRIM_INTS;
        const int kind = p[0], x0 = p[1], y0 = p[2], x1 = p[3], y1 = p[4];
        const unsigned col = (unsigned)p[5];
        if (kind == PRIM_MASK) {
          if (x < x0 || x >= x1 || y < y0 || y >= y1) continue;
          const unsigned m = a.masks[(long long)p[6] + (long long)(y - y0) * (x1 - x0) + (x - x0)];
          if (m == 0) continue;
          r = blend8(m, r, col & 255u); g = blend8(m, g, (col >> 8) & 255u); b = blend8(m, b, (col >> 16) & 255u);
          dirty = true;
          continue;
        }
        if (x < x0 || x > x1 || y < y0 || y > y1) continue;
        if (kind == PRIM_RING) {
          const int w = p[6];
          if (x >= x0 + w && x <= x1 - w && y >= y0 + w && y <= y1 - w) continue;      // inside the hole
        }
        r = col & 255u; g = (col >> 8) & 255u; b = (col >> 16) & 255u;
        dirty = true;
      }
    }
    __syncthreads();
  }
  if (live && dirty) { px[0] = (unsigned char)r; px[1] = (unsigned char)g; px[2] = (unsigned char)b; }
}

// ------------------------------------------------------------------------------------ PNG (stored deflate) + base64
constexpr unsigned CRC_POLY = 0xedb88320u;
constexpr int STORED_MAX = 65535;
constexpr int PNG_HEAD = 8 + 25 + 8;     // signature, IHDR chunk, IDAT length + type: first byte of the zlib stream
constexpr int CRC_SEG = 4096;            // bytes per CRC partial (all but the first segment, which takes the remainder)
constexpr unsigned ADLER_MOD = 65521u;

struct PngArgs {
  const unsigned char* img; unsigned char* png; unsigned* part; unsigned char* b64;
  int H, W;
  long long U, Z, total;       // uncompressed bytes, zlib stream bytes, file bytes
  int nblk, nseg;
  long long first_seg;         // length of CRC segment 0
};

__device__ __forceinline__ long long zpos(long long u) { return 2 + 5 * (u / STORED_MAX + 1) + u; }   // offset inside the zlib stream

// bit-at-a-time CRC step table entry (reflected polynomial)
__device__ __forceinline__ unsigned crc_entry(unsigned i) {
  unsigned c = i;
#pragma unroll
  for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
  return c;
}

__device__ __forceinline__ void put_be32(unsigned char* p, unsigned v) {
  p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v;
}

// scanlines (filter byte 0 + RGB) into the stored blocks; block headers, file header and trailer constants.
// One thread per 4 uncompressed bytes.
__global__ __launch_bounds__(256) void png_scatter_kernel(PngArgs a) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned char* z = a.png + PNG_HEAD;
  const int RB = 3 * a.W + 1;
  const long long u0 = t * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long long u = u0 + e;
    if (u >= a.U) break;
    const long long row = u / RB;
    const int col = (int)(u - row * RB);
    z[zpos(u)] = col == 0 ? (unsigned char)0 : a.img[row * (long long)(3 * a.W) + col - 1];
  }
  if (t < a.nblk) {                                    // stored-block header t
    const long long start = t * (long long)STORED_MAX;
    const unsigned len = (unsigned)min((long long)STORED_MAX, a.U - start);
    unsigned char* h = z + 2 + 5 * t + start;
    h[0] = t == a.nblk - 1 ? 1 : 0;
    h[1] = (unsigned char)(len & 255u); h[2] = (unsigned char)(len >> 8);
    h[3] = (unsigned char)(~len & 255u); h[4] = (unsigned char)((~len >> 8) & 255u);
  }
  if (t == 0) {
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    for (int k = 0; k < 8; ++k) a.png[k] = sig[k];
    unsigned char* ih = a.png + 8;
    put_be32(ih, 13); ih[4] = 'I'; ih[5] = 'H'; ih[6] = 'D'; ih[7] = 'R';
    put_be32(ih + 8, (unsigned)a.W); put_be32(ih + 12, (unsigned)a.H);
    ih[16] = 8; ih[17] = 2; ih[18] = 0; ih[19] = 0; ih[20] = 0;                 // 8 bits, truecolour, deflate, adaptive, no interlace
    unsigned c = 0xffffffffu;
    for (int k = 4; k < 21; ++k) c = crc_entry((c ^ ih[k]) & 255u) ^ (c >> 8);
    put_be32(ih + 21, c ^ 0xffffffffu);
    unsigned char* id = a.png + 33;
    put_be32(id, (unsigned)a.Z); id[4] = 'I'; id[5] = 'D'; id[6] = 'A'; id[7] = 'T';
    z[0] = 0x78; z[1] = 0x01;
    unsigned char* ie = a.png + PNG_HEAD + a.Z + 4;                              // behind the IDAT CRC
    const unsigned char iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xae, 0x42, 0x60, 0x82};
    for (int k = 0; k < 12; ++k) ie[k] = iend[k];
  }
}

// Adler-32 partials of the scanlines: one wave per row, part[2 row] = sum of bytes, part[2 row + 1] = sum of (RB - i) * byte_i
// (both mod 65521), i.e. the row's contribution to (A, B) when it is the LAST row.
__global__ __launch_bounds__(64) void png_adler_rows_kernel(PngArgs a) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const int RB = 3 * a.W + 1;
  const unsigned char* src = a.img + (long long)row * (3 * a.W);
  unsigned long long s1 = 0, s2 = 0;
  for (int i = 1 + lane; i < RB; i += 64) {            // byte 0 of the scanline is the filter byte 0: contributes nothing
    const unsigned v = src[i - 1];
    s1 += v;
    s2 += (unsigned long long)(RB - i) * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if (lane == 0) { a.part[2 * row] = (unsigned)(s1 % ADLER_MOD); a.part[2 * row + 1] = (unsigned)(s2 % ADLER_MOD); }
}

// one thread: fold the row partials into Adler-32 and store it (big endian) behind the last stored block
__global__ void png_adler_fold_kernel(PngArgs a) {
  if (threadIdx.x || blockIdx.x) return;
  const unsigned RB = (unsigned)(3 * a.W + 1);
  unsigned long long A = 1, B = 0;
  for (int r = 0; r < a.H; ++r) {
    // appending a row of RB bytes with byte sum s1 and weighted sum s2:  B += RB * A + s2,  A += s1
    B = (B + (unsigned long long)(RB % ADLER_MOD) * A + a.part[2 * r + 1]) % ADLER_MOD;
    A = (A + a.part[2 * r]) % ADLER_MOD;
  }
  put_be32(a.png + PNG_HEAD + a.Z - 4, (unsigned)((B << 16) | A));
}

// CRC-32 partials over "IDAT" + zlib stream: thread s covers segment s (segment 0 = the first `first_seg` bytes, the others
// CRC_SEG each); raw register value after running from 0 — the partials combine linearly (crc_fold).
__global__ __launch_bounds__(256) void png_crc_seg_kernel(PngArgs a) {
  __shared__ unsigned tab[256];
  tab[threadIdx.x] = crc_entry(threadIdx.x);
  __syncthreads();
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.nseg) return;
  const unsigned char* base = a.png + 37;                                        // the chunk type field
  const long long off = s == 0 ? 0 : a.first_seg + (long long)(s - 1) * CRC_SEG;
  const long long len = s == 0 ? a.first_seg : CRC_SEG;
  unsigned c = s == 0 ? 0xffffffffu : 0u;                                        // the initial value travels with segment 0
  for (long long i = 0; i < len; ++i) c = tab[(c ^ base[off + i]) & 255u] ^ (c >> 8);
  a.part[2 * a.H + s] = c;
}

__device__ unsigned gf2_mul(unsigned x, unsigned y) {      // product of two polynomials mod P, reflected bit order (zlib's multmodp)
  unsigned m = 1u << 31, p = 0;
  for (;;) {
    if (x & m) {
      p ^= y;
      if ((x & (m - 1)) == 0) break;
    }
    m >>= 1;
    y = (y & 1u) ? (y >> 1) ^ CRC_POLY : y >> 1;
  }
  return p;
}

// one thread: register after segment s = shift(register after segment s - 1, CRC_SEG bytes) ^ partial s, shift = multiply by
// x^(8 CRC_SEG) mod P; final xor; store behind the zlib stream
__global__ void png_crc_fold_kernel(PngArgs a) {
  if (threadIdx.x || blockIdx.x) return;
  unsigned xp = 1u << 30;                                  // x^1
  unsigned op = 1u << 31;                                  // x^0
  for (unsigned n = 8u * CRC_SEG; n; n >>= 1) {            // op = x^(8 CRC_SEG) by square and multiply
    if (n & 1u) op = gf2_mul(xp, op);
    xp = gf2_mul(xp, xp);
  }
  unsigned c = a.part[2 * a.H];
  for (int s = 1; s < a.nseg; ++s) c = gf2_mul(op, c) ^ a.part[2 * a.H + s];
  put_be32(a.png + PNG_HEAD + a.Z, c ^ 0xffffffffu);
}

// base64 of the file image: one thread per 3 input bytes
__global__ __launch_bounds__(256) void base64_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long long n) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i = g * 3;
  if (i >= n) return;
  const unsigned b0 = src[i], b1 = i + 1 < n ? src[i + 1] : 0u, b2 = i + 2 < n ? src[i + 2] : 0u;
  const unsigned v = (b0 << 16) | (b1 << 8) | b2;
  auto enc = [](unsigned s) -> unsigned char {
    return (unsigned char)(s < 26 ? 'A' + s : s < 52 ? 'a' + (s - 26) : s < 62 ? '0' + (s - 52) : s == 62 ? '+' : '/');
  };
  unsigned char* o = dst + g * 4;
  o[0] = enc((v >> 18) & 63u);
  o[1] = enc((v >> 12) & 63u);
  o[2] = i + 1 < n ? enc((v >> 6) & 63u) : (unsigned char)'=';
  o[3] = i + 2 < n ? enc(v & 63u) : (unsigned char)'=';
}

// ------------------------------------------------------------------------------------ PNG with a compressed stream
// (layout and rationale: oracle/png_ref.py — Up-filtered scanlines cut into 4096-byte units, one fixed-Huffman deflate block of
// literals + distance-1 / distance-3 run matches per unit, closed by an empty stored block so that units concatenate bytewise)
constexpr int UNIT = 4096;
constexpr int SLOT = 4640;               // >= UNIT * 9 / 8 + header / trailer bytes, 16-byte multiple
enum { M_Z = 0, M_TOTAL = 1, M_B64 = 2, M_NSEG = 3, M_SIZES = 4 };

struct DefArgs {
  const unsigned char* img; unsigned char* png; unsigned char* filt; unsigned char* slots; unsigned* meta; unsigned* part; unsigned char* b64;
  int H, W, nunits;
  long long U;
};

// filtered scanline stream: row 0 filter 0 (None), other rows filter 2 (Up).  One thread per 4 stream bytes.
__global__ __launch_bounds__(256) void png_filter_kernel(DefArgs a) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int RB = 3 * a.W + 1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long long u = t * 4 + e;
    if (u >= a.U) break;
    const long long row = u / RB;
    const int col = (int)(u - row * RB);
    unsigned v;
    if (col == 0) v = row ? 2u : 0u;
    else {
      const unsigned char* p = a.img + row * (long long)(3 * a.W) + col - 1;
      v = row ? (unsigned)p[0] - (unsigned)p[-3 * a.W] : (unsigned)p[0];
    }
    a.filt[u] = (unsigned char)v;
  }
}

struct BitWriter {
  unsigned char* out;
  unsigned long long acc;
  int nb, pos;
  __device__ __forceinline__ void put(unsigned v, int n) {        // LSB first
    acc |= (unsigned long long)v << nb;
    nb += n;
    while (nb >= 8) { out[pos++] = (unsigned char)(acc & 255u); acc >>= 8; nb -= 8; }
  }
  __device__ __forceinline__ void put_code(unsigned code, int n) {  // Huffman codes are packed starting from their MSB
    unsigned r = 0;
    for (int k = 0; k < n; ++k) { r = (r << 1) | (code & 1u); code >>= 1; }
    put(r, n);
  }
  __device__ __forceinline__ void lit_len(unsigned sym) {            // fixed code of RFC 1951 3.2.6
    if (sym < 144u) put_code(0x30u + sym, 8);
    else if (sym < 256u) put_code(0x190u + sym - 144u, 9);
    else if (sym < 280u) put_code(sym - 256u, 7);
    else put_code(0xC0u + sym - 280u, 8);
  }
  __device__ __forceinline__ void match(int len, int dist) {
    if (len <= 10) lit_len(257u + (unsigned)(len - 3));
    else if (len == 258) lit_len(285u);
    else {
      const unsigned m = (unsigned)(len - 3);
      const int e = 31 - __builtin_clz(m) - 2;
      const unsigned idx = (m >> e) - 4u;
      lit_len(257u + 4u + 4u * (unsigned)e + idx);
      put(m - ((idx + 4u) << e), e);
    }
    put_code(dist == 1 ? 0u : 2u, 5);
  }
};

// one thread per unit: greedy run matching (distance 1 = byte runs, distance 3 = pixel runs; matches may reach back across the unit
// start — the decoder's window holds those bytes), fixed-Huffman block + empty stored block; stored block if that is not smaller.
__global__ __launch_bounds__(64) void png_deflate_units_kernel(DefArgs a) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= a.nunits) return;
  const long long start = (long long)u * UNIT;
  const long long end = min(start + UNIT, a.U);
  const bool final_unit = u == a.nunits - 1;
  const unsigned char* f = a.filt;
  BitWriter b;
  b.out = a.slots + (long long)u * SLOT; b.acc = 0; b.nb = 0; b.pos = 0;
  b.put(0u, 1); b.put(1u, 2);
  long long i = start;
  while (i < end) {
    int l1 = 0, l3 = 0;
    if (i >= 1) { const unsigned char p = f[i - 1]; while (i + l1 < end && l1 < 258 && f[i + l1] == p) ++l1; }
    if (i >= 3) { while (i + l3 < end && l3 < 258 && f[i + l3] == f[i + l3 - 3]) ++l3; }
    const int best = l3 > l1 ? l3 : l1;
    if (best >= 3) { b.match(best, l3 > l1 ? 3 : 1); i += best; }
    else { b.lit_len(f[i]); ++i; }
  }
  b.lit_len(256u);
  b.put(final_unit ? 1u : 0u, 1); b.put(0u, 2);
  if (b.nb) { b.out[b.pos++] = (unsigned char)(b.acc & 255u); b.acc = 0; b.nb = 0; }
  b.out[b.pos++] = 0; b.out[b.pos++] = 0; b.out[b.pos++] = 0xff; b.out[b.pos++] = 0xff;
  const int n = (int)(end - start);
  if (b.pos >= n + 5) {
    unsigned char* o = b.out;
    o[0] = final_unit ? 1 : 0;
    o[1] = (unsigned char)(n & 255); o[2] = (unsigned char)(n >> 8);
    o[3] = (unsigned char)(~n & 255); o[4] = (unsigned char)((~n >> 8) & 255);
    for (int k = 0; k < n; ++k) o[5 + k] = f[start + k];
    b.pos = n + 5;
  }
  a.meta[M_SIZES + u] = (unsigned)b.pos;
}

// one thread: unit offsets (exclusive scan), stream / file sizes, constant header bytes
__global__ void png_layout_kernel(DefArgs a) {
  if (threadIdx.x || blockIdx.x) return;
  unsigned off = 0;
  for (int u = 0; u < a.nunits; ++u) {
    const unsigned sz = a.meta[M_SIZES + u];
    a.meta[M_SIZES + a.nunits + u] = off;
    off += sz;
  }
  const unsigned Z = 2u + off + 4u;
  a.meta[M_Z] = Z;
  a.meta[M_TOTAL] = Z + 57u;
  a.meta[M_B64] = 4u * ((Z + 57u + 2u) / 3u);
  a.meta[M_NSEG] = (4u + Z + CRC_SEG - 1u) / CRC_SEG;
  const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  for (int k = 0; k < 8; ++k) a.png[k] = sig[k];
  unsigned char* ih = a.png + 8;
  put_be32(ih, 13); ih[4] = 'I'; ih[5] = 'H'; ih[6] = 'D'; ih[7] = 'R';
  put_be32(ih + 8, (unsigned)a.W); put_be32(ih + 12, (unsigned)a.H);
  ih[16] = 8; ih[17] = 2; ih[18] = 0; ih[19] = 0; ih[20] = 0;
  unsigned c = 0xffffffffu;
  for (int k = 4; k < 21; ++k) c = crc_entry((c ^ ih[k]) & 255u) ^ (c >> 8);
  put_be32(ih + 21, c ^ 0xffffffffu);
  unsigned char* id = a.png + 33;
  put_be32(id, Z); id[4] = 'I'; id[5] = 'D'; id[6] = 'A'; id[7] = 'T';
  a.png[PNG_HEAD] = 0x78; a.png[PNG_HEAD + 1] = 0x01;
  unsigned char* ie = a.png + PNG_HEAD + Z + 4;
  const unsigned char iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xae, 0x42, 0x60, 0x82};
  for (int k = 0; k < 12; ++k) ie[k] = iend[k];
}

// one wave per unit: slot -> its place in the stream
__global__ __launch_bounds__(64) void png_gather_kernel(DefArgs a) {
  const int u = blockIdx.x;
  const unsigned sz = a.meta[M_SIZES + u], off = a.meta[M_SIZES + a.nunits + u];
  const unsigned char* src = a.slots + (long long)u * SLOT;
  unsigned char* dst = a.png + PNG_HEAD + 2 + off;
  for (unsigned k = threadIdx.x; k < sz; k += 64) dst[k] = src[k];
}


// ------------------------------------------------------------------------------------ PNG with an LZ77 + dynamic-Huffman stream
// OMNI_OP_PNG_DEFLATE i5 = 1 (round 6).  Specification, step by step and tie-break by tie-break: oracle/png_ref.py::deflate_unit_lz —
// this kernel must produce those bytes.  One 32 KiB unit of the Up-filtered stream per workgroup; LANE 0 does the (inherently serial)
// greedy tokenisation with a 13-bit, one-entry-per-bucket hash table in LDS (plus the distance-1 / distance-3 run candidates of the
// fixed-Huffman variant), counts symbol frequencies on the way, builds the two length-limited Huffman codes and the code-length code
// in LDS, and writes ONE dynamic block + an empty stored block into the unit's slot (stored form if that is not smaller).  190 units
// per 1080p frame: a latency-bound kernel by construction (one lane per CU's worth of work), ~10 ms — against ~100-300 ms of host
// zlib for the reference's PNG (ref:util/utils.py:485-488) and 2.7x the bytes for the fixed-Huffman variant above.
constexpr int UNIT_LZ = 32768;
constexpr int SLOT_LZ = UNIT_LZ + 1024;   // a short unit's block header (<= ~600 bytes) is written before its size is known
constexpr int HASH_BITS = 13;
constexpr int NLIT = 286, NDIST = 30, NCL = 19;

struct LzWork {
  unsigned short table[1 << HASH_BITS];
  unsigned freq[NLIT + NDIST + NCL];
  unsigned char lens[NLIT + NDIST + NCL];
  unsigned short codes[NLIT + NDIST + NCL];
  unsigned weight[2 * NLIT - 1];
  unsigned short parent[2 * NLIT - 1];
  unsigned short depth[2 * NLIT - 1];
  unsigned short order[NLIT];
  unsigned rle[NLIT + NDIST];            // symbol | extra bit count << 8 | extra value << 16
};

__device__ __forceinline__ unsigned lz_hash3(unsigned b0, unsigned b1, unsigned b2) {
  return (((b0 << 16) | (b1 << 8) | b2) * 0x9E3779B1u) >> (32 - HASH_BITS);
}
// length symbol index k (0..28), extra bits, extra value of a match length 3..258 (RFC 1951 3.2.5)
__device__ __forceinline__ void lz_len_code(int len, int& k, int& eb, unsigned& ev) {
  if (len <= 10) { k = len - 3; eb = 0; ev = 0; }
  else if (len == 258) { k = 28; eb = 0; ev = 0; }
  else {
    const unsigned m = (unsigned)(len - 3);
    const int e = 31 - __builtin_clz(m) - 2;
    const unsigned idx = (m >> e) - 4u;
    k = 4 + 4 * e + (int)idx; eb = e; ev = m - ((idx + 4u) << e);
  }
}
__device__ __forceinline__ void lz_dist_code(int dist, int& k, int& eb, unsigned& ev) {
  if (dist <= 4) { k = dist - 1; eb = 0; ev = 0; }
  else {
    const unsigned m = (unsigned)(dist - 1);
    const int e = 31 - __builtin_clz(m) - 1;
    k = 2 * (e + 1) + (int)((m >> e) & 1u); eb = e; ev = m & ((1u << e) - 1u);
  }
}

// oracle/png_ref.py::huffman_lengths: two-queue Huffman over the used symbols sorted by (frequency, symbol), Kraft repair to max_bits
__device__ void lz_huffman_lengths(LzWork& w, const unsigned* freq, int n, int max_bits, unsigned char* lens) {
  int m = 0;
  for (int s = 0; s < n; ++s) { lens[s] = 0; if (freq[s] > 0) w.order[m++] = (unsigned short)s; }
  for (int i = 1; i < m; ++i) {                        // stable insertion sort: equal frequencies stay in symbol order
    const unsigned short key = w.order[i];
    const unsigned kf = freq[key];
    int j = i - 1;
    while (j >= 0 && freq[w.order[j]] > kf) { w.order[j + 1] = w.order[j]; --j; }
    w.order[j + 1] = key;
  }
  if (m == 0) return;
  if (m == 1) { lens[w.order[0]] = 1; return; }
  for (int k = 0; k < m; ++k) w.weight[k] = freq[w.order[k]];
  int leaf = 0, inode = m, nxt = m;
  for (int it = 0; it < m - 1; ++it) {
    int pick[2];
    for (int k = 0; k < 2; ++k) {
      if (leaf < m && (inode >= nxt || w.weight[leaf] <= w.weight[inode])) pick[k] = leaf++;
      else pick[k] = inode++;
    }
    w.weight[nxt] = w.weight[pick[0]] + w.weight[pick[1]];
    w.parent[pick[0]] = (unsigned short)nxt; w.parent[pick[1]] = (unsigned short)nxt;
    ++nxt;
  }
  w.depth[2 * m - 2] = 0;
  for (int k = 2 * m - 3; k >= 0; --k) w.depth[k] = (unsigned short)(w.depth[w.parent[k]] + 1);
  int count[16];
  for (int b = 0; b <= max_bits; ++b) count[b] = 0;
  for (int k = 0; k < m; ++k) count[min((int)w.depth[k], max_bits)]++;
  long long total = 0;
  for (int b = 1; b <= max_bits; ++b) total += (long long)count[b] << (max_bits - b);
  while (total > (1ll << max_bits)) {
    count[max_bits]--;
    for (int b = max_bits - 1; b > 0; --b)
      if (count[b]) { count[b]--; count[b + 1] += 2; break; }
    --total;
  }
  int k = 0;
  for (int b = max_bits; b > 0; --b)
    for (int c = 0; c < count[b]; ++c) lens[w.order[k++]] = (unsigned char)b;
}

__device__ void lz_canonical_codes(const unsigned char* lens, int n, int max_bits, unsigned short* codes) {
  int bl[17], nxt[17];
  for (int b = 0; b <= max_bits + 1; ++b) bl[b] = 0;
  for (int s = 0; s < n; ++s) bl[lens[s]]++;
  bl[0] = 0;
  int code = 0;
  for (int b = 1; b <= max_bits; ++b) { code = (code + bl[b - 1]) << 1; nxt[b] = code; }
  for (int s = 0; s < n; ++s) { codes[s] = 0; if (lens[s]) codes[s] = (unsigned short)nxt[lens[s]]++; }
}

__global__ __launch_bounds__(64) void png_lz_units_kernel(DefArgs a, unsigned* __restrict__ toks_all) {
  __shared__ LzWork w;
  if (threadIdx.x != 0) return;                          // one lane per unit (see the header comment)
  const int u = blockIdx.x;
  const long long start = (long long)u * UNIT_LZ;
  const long long end = min(start + (long long)UNIT_LZ, a.U);
  const bool final_unit = u == a.nunits - 1;
  const unsigned char* __restrict__ f = a.filt;
  unsigned* __restrict__ toks = toks_all + (long long)u * UNIT_LZ;
  for (int k = 0; k < (1 << HASH_BITS); ++k) w.table[k] = 0;
  unsigned* lf = w.freq; unsigned* df = w.freq + NLIT; unsigned* cf = w.freq + NLIT + NDIST;
  for (int k = 0; k < NLIT + NDIST + NCL; ++k) w.freq[k] = 0;
  // ---- tokens
  int nt = 0;
  auto mlen = [&](long long i, long long d) {
    int l = 0;
    while (i + l < end && l < 258 && f[i + l] == f[i + l - d]) ++l;
    return l;
  };
  auto insert = [&](long long q) {
    if (q + 2 < end) w.table[lz_hash3(f[q], f[q + 1], f[q + 2])] = (unsigned short)(q - start + 1);
  };
  long long i = start;
  while (i < end) {
    int best = 0; long long bd = 0;
    if (i >= 1) { best = mlen(i, 1); bd = 1; }
    if (i >= 3) { const int l = mlen(i, 3); if (l > best) { best = l; bd = 3; } }
    if (i + 2 < end) {
      const unsigned c = w.table[lz_hash3(f[i], f[i + 1], f[i + 2])];
      if (c) {
        const long long d = i - (start + (long long)c - 1);
        int l = mlen(i, d);
        if (d > 3 && l < 4) l = 0;
        if (l > best || (l == best && l > 0 && d < bd)) { best = l; bd = d; }
      }
    }
    if (best >= 3) {
      toks[nt++] = ((unsigned)best << 16) | (unsigned)bd;
      int k, eb; unsigned ev;
      lz_len_code(best, k, eb, ev); lf[257 + k]++;
      lz_dist_code((int)bd, k, eb, ev); df[k]++;
      for (long long q = i; q < i + best; ++q) insert(q);
      i += best;
    } else {
      toks[nt++] = (unsigned)f[i] << 16;
      lf[f[i]]++;
      insert(i);
      ++i;
    }
  }
  lf[256]++;
  {
    int used = 0;
    for (int s = 0; s < NLIT; ++s) used += lf[s] != 0;
    if (used < 2) lf[lf[0] == 0 ? 0 : 1]++;
    for (int s = 0; s < 2; ++s) {
      used = 0;
      for (int t = 0; t < NDIST; ++t) used += df[t] != 0;
      if (used < 2 && df[s] == 0) df[s] = 1;
    }
  }
  unsigned char* ll = w.lens; unsigned char* dl = w.lens + NLIT; unsigned char* cl = w.lens + NLIT + NDIST;
  unsigned short* lc = w.codes; unsigned short* dc = w.codes + NLIT; unsigned short* cc = w.codes + NLIT + NDIST;
  lz_huffman_lengths(w, lf, NLIT, 15, ll);
  lz_huffman_lengths(w, df, NDIST, 15, dl);
  lz_canonical_codes(ll, NLIT, 15, lc);
  lz_canonical_codes(dl, NDIST, 15, dc);
  int hlit = 257, hdist = 1;
  for (int s = 0; s < NLIT; ++s) if (ll[s]) hlit = max(hlit, s + 1);
  for (int s = 0; s < NDIST; ++s) if (dl[s]) hdist = max(hdist, s + 1);
  // ---- run-length coding of the code-length sequence ll[0..hlit) ++ dl[0..hdist)
  int nr = 0;
  {
    const int n = hlit + hdist;
    auto at = [&](int k) -> int { return k < hlit ? ll[k] : dl[k - hlit]; };
    int p = 0;
    while (p < n) {
      const int v = at(p);
      int j = p;
      while (j < n && at(j) == v) ++j;
      int run = j - p;
      if (v == 0) {
        while (run >= 11) { const int r = min(run, 138); w.rle[nr++] = 18u | (7u << 8) | ((unsigned)(r - 11) << 16); run -= r; }
        if (run >= 3) { w.rle[nr++] = 17u | (3u << 8) | ((unsigned)(run - 3) << 16); run = 0; }
        for (; run > 0; --run) w.rle[nr++] = 0u;
      } else {
        w.rle[nr++] = (unsigned)v; --run;
        while (run >= 3) { const int r = min(run, 6); w.rle[nr++] = 16u | (2u << 8) | ((unsigned)(r - 3) << 16); run -= r; }
        for (; run > 0; --run) w.rle[nr++] = (unsigned)v;
      }
      p = j;
    }
  }
  for (int k = 0; k < nr; ++k) cf[w.rle[k] & 255u]++;
  {
    const int force[2] = {0, 18};
    for (int t = 0; t < 2; ++t) {
      int used = 0;
      for (int s = 0; s < NCL; ++s) used += cf[s] != 0;
      if (used < 2 && cf[force[t]] == 0) cf[force[t]] = 1;
    }
  }
  lz_huffman_lengths(w, cf, NCL, 7, cl);
  lz_canonical_codes(cl, NCL, 7, cc);
  const int CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 4;
  for (int k = 0; k < 19; ++k) if (cl[CL_ORDER[k]]) hclen = max(hclen, k + 1);
  // ---- the block
  BitWriter b;
  b.out = a.slots + (long long)u * SLOT_LZ; b.acc = 0; b.nb = 0; b.pos = 0;
  const int n = (int)(end - start);
  const int limit = n + 5;                               // beyond this the stored form wins: stop writing (the slot holds n + 1024 bytes)
  b.put(0u, 1); b.put(2u, 2);
  b.put((unsigned)(hlit - 257), 5); b.put((unsigned)(hdist - 1), 5); b.put((unsigned)(hclen - 4), 4);
  for (int k = 0; k < hclen; ++k) b.put(cl[CL_ORDER[k]], 3);
  for (int k = 0; k < nr; ++k) {
    const unsigned r = w.rle[k], s = r & 255u, eb = (r >> 8) & 255u;
    b.put_code(cc[s], cl[s]);
    if (eb) b.put(r >> 16, (int)eb);
  }
  bool overflow = false;
  for (int t = 0; t < nt; ++t) {
    if (b.pos > limit) { overflow = true; break; }
    const unsigned tk = toks[t];
    const int len = (int)(tk >> 16), dist = (int)(tk & 0xffffu);
    if (dist == 0) b.put_code(lc[len], ll[len]);
    else {
      int k, eb; unsigned ev;
      lz_len_code(len, k, eb, ev);
      b.put_code(lc[257 + k], ll[257 + k]);
      if (eb) b.put(ev, eb);
      lz_dist_code(dist, k, eb, ev);
      b.put_code(dc[k], dl[k]);
      if (eb) b.put(ev, eb);
    }
  }
  if (!overflow) {
    b.put_code(lc[256], ll[256]);
    b.put(final_unit ? 1u : 0u, 1); b.put(0u, 2);
    if (b.nb) { b.out[b.pos++] = (unsigned char)(b.acc & 255u); b.acc = 0; b.nb = 0; }
    b.out[b.pos++] = 0; b.out[b.pos++] = 0; b.out[b.pos++] = 0xff; b.out[b.pos++] = 0xff;
  }
  if (overflow || b.pos >= n + 5) {
    unsigned char* o = b.out;
    o[0] = final_unit ? 1 : 0;
    o[1] = (unsigned char)(n & 255); o[2] = (unsigned char)(n >> 8);
    o[3] = (unsigned char)(~n & 255); o[4] = (unsigned char)((~n >> 8) & 255);
    for (int k = 0; k < n; ++k) o[5 + k] = f[start + k];
    b.pos = n + 5;
  }
  a.meta[M_SIZES + u] = (unsigned)b.pos;
}

// one wave per unit: LZ slot -> its place in the stream
__global__ __launch_bounds__(64) void png_lz_gather_kernel(DefArgs a) {
  const int u = blockIdx.x;
  const unsigned sz = a.meta[M_SIZES + u], off = a.meta[M_SIZES + a.nunits + u];
  const unsigned char* src = a.slots + (long long)u * SLOT_LZ;
  unsigned char* dst = a.png + PNG_HEAD + 2 + off;
  for (unsigned k = threadIdx.x; k < sz; k += 64) dst[k] = src[k];
}

// Adler-32 partials over the rows of the FILTERED stream (filter byte included), same form as png_adler_rows_kernel
__global__ __launch_bounds__(64) void png_adler_filt_kernel(DefArgs a) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const int RB = 3 * a.W + 1;
  const unsigned char* src = a.filt + (long long)row * RB;
  unsigned long long s1 = 0, s2 = 0;
  for (int i = lane; i < RB; i += 64) {
    const unsigned v = src[i];
    s1 += v;
    s2 += (unsigned long long)(RB - i) * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if (lane == 0) { a.part[2 * row] = (unsigned)(s1 % ADLER_MOD); a.part[2 * row + 1] = (unsigned)(s2 % ADLER_MOD); }
}

// the fold / CRC kernels of the stored variant, with the stream length read from the device
__global__ void png_def_adler_fold_kernel(DefArgs a) {
  if (threadIdx.x || blockIdx.x) return;
  const unsigned RB = (unsigned)(3 * a.W + 1);
  unsigned long long A = 1, B = 0;
  for (int r = 0; r < a.H; ++r) {
    B = (B + (unsigned long long)(RB % ADLER_MOD) * A + a.part[2 * r + 1]) % ADLER_MOD;
    A = (A + a.part[2 * r]) % ADLER_MOD;
  }
  put_be32(a.png + PNG_HEAD + a.meta[M_Z] - 4, (unsigned)((B << 16) | A));
}

__global__ __launch_bounds__(256) void png_def_crc_seg_kernel(DefArgs a) {
  __shared__ unsigned tab[256];
  tab[threadIdx.x] = crc_entry(threadIdx.x);
  __syncthreads();
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int nseg = (int)a.meta[M_NSEG];
  if (s >= nseg) return;
  const long long covered = 4ll + a.meta[M_Z];
  const long long first = covered - (long long)(nseg - 1) * CRC_SEG;
  const unsigned char* base = a.png + 37;
  const long long off = s == 0 ? 0 : first + (long long)(s - 1) * CRC_SEG;
  const long long len = s == 0 ? first : CRC_SEG;
  unsigned c = s == 0 ? 0xffffffffu : 0u;
  for (long long i = 0; i < len; ++i) c = tab[(c ^ base[off + i]) & 255u] ^ (c >> 8);
  a.part[2 * a.H + s] = c;
}

__global__ void png_def_crc_fold_kernel(DefArgs a) {
  if (threadIdx.x || blockIdx.x) return;
  unsigned xp = 1u << 30, op = 1u << 31;
  for (unsigned n = 8u * CRC_SEG; n; n >>= 1) {
    if (n & 1u) op = gf2_mul(xp, op);
    xp = gf2_mul(xp, xp);
  }
  const int nseg = (int)a.meta[M_NSEG];
  unsigned c = a.part[2 * a.H];
  for (int s = 1; s < nseg; ++s) c = gf2_mul(op, c) ^ a.part[2 * a.H + s];
  put_be32(a.png + PNG_HEAD + a.meta[M_Z], c ^ 0xffffffffu);
}

__global__ __launch_bounds__(256) void base64_dyn_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                         const unsigned* __restrict__ meta) {
  const long long n = meta[M_TOTAL];
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i = g * 3;
  if (i >= n) return;
  const unsigned b0 = src[i], b1 = i + 1 < n ? src[i + 1] : 0u, b2 = i + 2 < n ? src[i + 2] : 0u;
  const unsigned v = (b0 << 16) | (b1 << 8) | b2;
  auto enc = [](unsigned s) -> unsigned char {
    return (unsigned char)(s < 26 ? 'A' + s : s < 52 ? 'a' + (s - 26) : s < 62 ? '0' + (s - 52) : s == 62 ? '+' : '/');
  };
  unsigned char* o = dst + g * 4;
  o[0] = enc((v >> 18) & 63u);
  o[1] = enc((v >> 12) & 63u);
  o[2] = i + 1 < n ? enc((v >> 6) & 63u) : (unsigned char)'=';
  o[3] = i + 2 < n ? enc(v & 63u) : (unsigned char)'=';
}

}  // namespace

// OMNI_OP_OVERLAY (see include/omni_amd.h)
int omni_launch_overlay(const omni_op_t* op, hipStream_t s) {
  OvArgs a{};
  a.img = (unsigned char*)op->p[0]; a.prim = (const int*)op->p[1]; a.masks = (const unsigned char*)op->p[2];
  a.H = op->i[0]; a.W = op->i[1]; a.n = op->i[2];
  OMNI_REQUIRE(a.img && a.H > 0 && a.W > 0 && a.n >= 0 && (a.n == 0 || a.prim), "overlay: bad arguments");
  if (a.n == 0) return OMNI_OK;
  hipLaunchKernelGGL(overlay_kernel, dim3((a.W + 63) / 64, (a.H + 3) / 4), dim3(256), 0, s, a);
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

// OMNI_OP_PNG_PACK (see include/omni_amd.h)
int omni_launch_png_pack(const omni_op_t* op, hipStream_t s) {
  PngArgs a{};
  a.img = (const unsigned char*)op->p[0]; a.png = (unsigned char*)op->p[1]; a.part = (unsigned*)op->p[2]; a.b64 = (unsigned char*)op->p[3];
  a.H = op->i[0]; a.W = op->i[1];
  OMNI_REQUIRE(a.img && a.png && a.part && a.H > 0 && a.W > 0 && a.H <= 32768 && a.W <= 32768, "png_pack: bad arguments");
  a.U = (long long)a.H * (3 * a.W + 1);
  a.nblk = (int)((a.U + STORED_MAX - 1) / STORED_MAX);
  a.Z = 2 + 5ll * a.nblk + a.U + 4;
  a.total = a.Z + 57;
  OMNI_REQUIRE(a.Z < (1ll << 31), "png_pack: image too large for one IDAT chunk");
  const long long covered = 4 + a.Z;                      // chunk type + data
  a.nseg = (int)((covered + CRC_SEG - 1) / CRC_SEG);
  a.first_seg = covered - (long long)(a.nseg - 1) * CRC_SEG;
  OMNI_REQUIRE(op->i[2] >= 2 * a.H + a.nseg, "png_pack: scratch holds %d words, needs %d", op->i[2], 2 * a.H + a.nseg);
  OMNI_REQUIRE(op->i[3] == 0 || op->i[3] >= a.total, "png_pack: output holds %d bytes, needs %lld", op->i[3], a.total);
  const long long quads = (a.U + 3) / 4;
  hipLaunchKernelGGL(png_scatter_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(png_adler_rows_kernel, dim3(a.H), dim3(64), 0, s, a);
  hipLaunchKernelGGL(png_adler_fold_kernel, dim3(1), dim3(64), 0, s, a);
  hipLaunchKernelGGL(png_crc_seg_kernel, dim3((a.nseg + 255) / 256), dim3(256), 0, s, a);
  hipLaunchKernelGGL(png_crc_fold_kernel, dim3(1), dim3(64), 0, s, a);
  if (a.b64) {
    const long long groups = (a.total + 2) / 3;
    hipLaunchKernelGGL(base64_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, (const unsigned char*)a.png, a.b64, a.total);
  }
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

// OMNI_OP_PNG_DEFLATE (see include/omni_amd.h)
int omni_launch_png_deflate(const omni_op_t* op, hipStream_t s) {
  DefArgs a{};
  a.img = (const unsigned char*)op->p[0]; a.png = (unsigned char*)op->p[1]; a.filt = (unsigned char*)op->p[2];
  a.slots = (unsigned char*)op->p[3]; a.meta = (unsigned*)op->p[4]; a.part = (unsigned*)op->p[5]; a.b64 = (unsigned char*)op->p[6];
  a.H = op->i[0]; a.W = op->i[1];
  OMNI_REQUIRE(a.img && a.png && a.filt && a.slots && a.meta && a.part && a.H > 0 && a.W > 0 && a.H <= 32768 && a.W <= 32768,
               "png_deflate: bad arguments");
  a.U = (long long)a.H * (3 * a.W + 1);
  const bool lz = op->i[5] == 1;                          // LZ77 + dynamic Huffman over 32 KiB units (p7 = token scratch, u32[units * 32768])
  unsigned* toks = (unsigned*)op->p[7];
  OMNI_REQUIRE(!lz || toks, "png_deflate: i5 = 1 needs the token scratch p7");
  a.nunits = lz ? (int)((a.U + UNIT_LZ - 1) / UNIT_LZ) : (int)((a.U + UNIT - 1) / UNIT);
  const long long zmax = 2 + a.U + 5ll * ((a.U + UNIT - 1) / UNIT) + 4;     // capacity contract of the fixed-Huffman variant (covers both)
  OMNI_REQUIRE(zmax < (1ll << 31), "png_deflate: image too large for one IDAT chunk");
  const int nseg_max = (int)((4 + zmax + CRC_SEG - 1) / CRC_SEG);
  OMNI_REQUIRE(op->i[2] >= zmax + 57, "png_deflate: output holds %d bytes, worst case is %lld", op->i[2], zmax + 57);
  OMNI_REQUIRE(op->i[3] >= M_SIZES + 2 * a.nunits, "png_deflate: meta holds %d words, needs %d", op->i[3], M_SIZES + 2 * a.nunits);
  OMNI_REQUIRE(op->i[4] >= 2 * a.H + nseg_max, "png_deflate: scratch holds %d words, needs %d", op->i[4], 2 * a.H + nseg_max);
  const long long quads = (a.U + 3) / 4;
  hipLaunchKernelGGL(png_filter_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a);
  if (lz) hipLaunchKernelGGL(png_lz_units_kernel, dim3(a.nunits), dim3(64), 0, s, a, toks);
  else hipLaunchKernelGGL(png_deflate_units_kernel, dim3((a.nunits + 63) / 64), dim3(64), 0, s, a);
  hipLaunchKernelGGL(png_layout_kernel, dim3(1), dim3(64), 0, s, a);
  if (lz) hipLaunchKernelGGL(png_lz_gather_kernel, dim3(a.nunits), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(png_gather_kernel, dim3(a.nunits), dim3(64), 0, s, a);
  hipLaunchKernelGGL(png_adler_filt_kernel, dim3(a.H), dim3(64), 0, s, a);
  hipLaunchKernelGGL(png_def_adler_fold_kernel, dim3(1), dim3(64), 0, s, a);
  hipLaunchKernelGGL(png_def_crc_seg_kernel, dim3((nseg_max + 255) / 256), dim3(256), 0, s, a);
  hipLaunchKernelGGL(png_def_crc_fold_kernel, dim3(1), dim3(64), 0, s, a);
  if (a.b64) {
    const long long groups = (zmax + 57 + 2) / 3;
    hipLaunchKernelGGL(base64_dyn_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, (const unsigned char*)a.png, a.b64,
                       (const unsigned*)a.meta);
  }
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}
