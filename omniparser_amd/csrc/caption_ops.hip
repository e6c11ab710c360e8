// Florence-2 (DaViT + BART) kernels that are not GEMMs, for gfx950.  Token tensors are [rows, C]
// (NHWC with the spatial dims flattened); linear layers go through conv_igemm.hip.
//
//   dwconv3        x + depthwise3x3(x) + bias           hf:models/florence2/modeling_florence2.py:432-436,296-300
//   layernorm      nn.LayerNorm (optionally of x + table[row % period])   :154,281,292,417,428,574; bart:272-341
//   attn_rows      softmax(q k^T * scale) v, one thread per query row, K/V tiles broadcast from LDS
//                  mode 0: plain MHA (BART encoder)     hf:models/bart/modeling_bart.py:143-257
//                  mode 1: DaViT 12x12 window attention with UNMASKED zero-padded windows (:338-398)
//   chan_attn      DaViT grouped channel attention (:223-259): 32x32 score matrix per (image, group)
//   proj_prep      + learned 2-D position + sinusoid(t=0), [mean token ; tokens]   (:568-590, 56-113)
//   assemble       encoder input = [image features ; prompt embeddings]            (:933-960)
//   embed_step     decoder token embedding + learned position (offset 2)           bart:80-98,594-640
//   attn_decode    single-query attention with KV-cache append (self) / fixed K,V (cross)
//   greedy_step    final_logits_bias + NoRepeatNGram + ForcedBOS/EOS + argmax + EOS/pad bookkeeping
//                  hf:generation/utils.py:2783-2937, hf:generation/logits_process.py:1115-1139,1556,1601
//   crop_resize    crop -> cv2.resize(64x64, INTER_LINEAR) -> [PIL BICUBIC to RxR] -> /255, normalise
//                  ref:util/utils.py:97-105,120-123 + hf CLIP image processor
// All softmax / LayerNorm statistics are f32; f16 tensors are converted on load.
#include "omni_internal.h"
#include <stdlib.h>

#pragma clang fp contract(off)

OMNI_RANGE_GUARD_TU()

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return ElemTraits<T>::to_f32(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = ElemTraits<T>::from_f32(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ------------------------------------------------------------------------------------ dwconv3
struct DwArgs { const void* x; const void* w; const float* bias; void* y; int B, H, W, C; long long total; };

// Per-output body of the original kernel (one 16-byte channel vector of one pixel).  Kept as the fallback for channel counts whose
// vector count per pixel is not a power of two, and as the bit-exact reference of the strip kernel below.
template <typename T>
__host__ __device__ __forceinline__ void dwconv3_point_body(const DwArgs& a, long long idx) {
  constexpr int V = ElemTraits<T>::kVec;
  struct Vec { T v[V]; };
  const T* __restrict__ X = (const T*)a.x;
  const T* __restrict__ Wt = (const T*)a.w;
  T* __restrict__ Y = (T*)a.y;
  const int cv = a.C / V;
  int c = (int)(idx % cv) * V;
  long long pix = idx / cv;
  int w = (int)(pix % a.W);
  long long t = pix / a.W;
  int h = (int)(t % a.H);
  long long b = t / a.H;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.0f;
  Vec ctr;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    int hi = h + r - 1;
    if (hi < 0 || hi >= a.H) continue;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      int wi = w + s - 1;
      if (wi < 0 || wi >= a.W) continue;
      Vec xv = __builtin_bit_cast(Vec, *reinterpret_cast<const u32x4*>(X + ((b * a.H + hi) * a.W + wi) * a.C + c));
      Vec wv = __builtin_bit_cast(Vec, *reinterpret_cast<const u32x4*>(Wt + (r * 3 + s) * a.C + c));
      if (r == 1 && s == 1) ctr = xv;
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += ElemTraits<T>::to_f32(wv.v[e]) * ElemTraits<T>::to_f32(xv.v[e]);
    }
  }
  Vec out;
#pragma unroll
  for (int e = 0; e < V; ++e) out.v[e] = ElemTraits<T>::from_f32((acc[e] + a.bias[c + e]) + ElemTraits<T>::to_f32(ctr.v[e]));
  *reinterpret_cast<u32x4*>(Y + pix * a.C + c) = __builtin_bit_cast(u32x4, out);
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv3_kernel(DwArgs a) {
  // w layout: [3][3][C] (tap-major) in T; y = x + bias + sum_taps w*x   (conv(x) + x).
  // One 16-byte channel vector per lane: consecutive lanes walk consecutive channels of a pixel, the
  // 3x3 neighbourhood is re-read through L1/L2, HBM sees x once and y once.
  const long long total = a.total / ElemTraits<T>::kVec;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x)
    dwconv3_point_body<T>(a, idx);
}

// Strip version (default when C / vector width is a power of two, i.e. every DaViT stage): the round-1 kernel above
// spends ~800 instructions per output vector, most of them 64-bit div/mod index arithmetic (static count:
// tools/isa_report-style dump, 461 VALU + 347 SALU for 72 useful mul/add) — it was issue-bound at ~2.3 TB/s, not
// memory-bound.  Here the grid is (row chunk, strip of DW_ROWS output rows, image): no division at all (w and c come
// from shifts), the 9 weight vectors and the bias are loaded once per thread, and each thread slides down DW_ROWS
// output rows so an input row is loaded once per 3 taps instead of 3 times ((R+2)/R loads per output row instead
// of 3).  The accumulation order per output element is exactly the original's (tap rows ascending, then columns),
// so results are bit-identical to dwconv3_kernel — checked on the host by tests/test_host_cpu.py.
constexpr int DW_ROWS = 4;

template <typename T>
__host__ __device__ __forceinline__ void dwconv3_strip_body(const DwArgs& a, int b, int h0, unsigned r, int cv_log2) {
  constexpr int V = ElemTraits<T>::kVec;
  struct Vec { T v[V]; };
  const int w = (int)(r >> cv_log2);
  if (w >= a.W) return;
  const int c = (int)(r & ((1u << cv_log2) - 1u)) * V;
  const T* __restrict__ X = (const T*)a.x + (long long)b * a.H * a.W * a.C + c;
  const T* __restrict__ Wt = (const T*)a.w + c;
  T* __restrict__ Y = (T*)a.y + (long long)b * a.H * a.W * a.C + c;
  float wt[9][V];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    Vec wv = __builtin_bit_cast(Vec, *reinterpret_cast<const u32x4*>(Wt + k * a.C));
#pragma unroll
    for (int e = 0; e < V; ++e) wt[k][e] = ElemTraits<T>::to_f32(wv.v[e]);
  }
  float acc[DW_ROWS][V];
  Vec ctr[DW_ROWS];
#pragma unroll
  for (int o = 0; o < DW_ROWS; ++o)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[o][e] = 0.0f;
#pragma unroll
  for (int i = 0; i < DW_ROWS + 2; ++i) {            // input rows h0-1 .. h0+DW_ROWS
    const int hi = h0 + i - 1;
    if (hi < 0 || hi >= a.H) continue;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int wi = w + s - 1;
      if (wi < 0 || wi >= a.W) continue;
      Vec xv = __builtin_bit_cast(Vec, *reinterpret_cast<const u32x4*>(X + ((long long)hi * a.W + wi) * a.C));
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {               // tap row rr of output row hi - rr + 1
        const int o = i - rr;                        // output index within the strip
        if (o < 0 || o >= DW_ROWS) continue;
        if (rr == 1 && s == 1) ctr[o] = xv;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[o][e] += wt[rr * 3 + s][e] * ElemTraits<T>::to_f32(xv.v[e]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < DW_ROWS; ++o) {
    const int ho = h0 + o;
    if (ho >= a.H) break;
    Vec out;
#pragma unroll
    for (int e = 0; e < V; ++e) out.v[e] = ElemTraits<T>::from_f32((acc[o][e] + a.bias[c + e]) + ElemTraits<T>::to_f32(ctr[o].v[e]));
    *reinterpret_cast<u32x4*>(Y + ((long long)ho * a.W + w) * a.C) = __builtin_bit_cast(u32x4, out);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv3_strip_kernel(DwArgs a, int cv_log2) {
  dwconv3_strip_body<T>(a, (int)blockIdx.z, (int)blockIdx.y * DW_ROWS, blockIdx.x * 256u + threadIdx.x, cv_log2);
}

// launch geometry shared by the device launcher and the host emulation
struct DwStripGrid { unsigned gx, gy, gz; int cv_log2; bool ok; };
inline DwStripGrid dwconv3_strip_grid(const DwArgs& a, int V) {
  DwStripGrid g{0, 0, 0, 0, false};
  const int cv = a.C / V;
  if (cv <= 0 || (cv & (cv - 1))) return g;                                   // power-of-two vector count only
  while ((1 << g.cv_log2) < cv) ++g.cv_log2;
  const long long row_vecs = (long long)a.W * cv;
  g.gx = (unsigned)((row_vecs + 255) / 256);
  g.gy = (unsigned)((a.H + DW_ROWS - 1) / DW_ROWS);
  g.gz = (unsigned)a.B;
  g.ok = g.gy <= 65535u && g.gz <= 65535u && row_vecs < (1ll << 31);
  return g;
}

// ------------------------------------------------------------------------------------ layernorm
struct LnArgs { const void* x; const void* add; const float* g; const float* b; void* y; void* y2; long long rows; int C, period, omode; float eps; };

template <typename T, int NIT>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
  // one wave per row; two-pass mean / variance in f32 (row cached in registers, C <= 64*NIT).  NIT is the number of
  // 64-channel slices actually needed (2/4/8/12/16): the round-1 kernel always ran 16 predicated slices, i.e. 8x the
  // instructions a C = 128 row needs (it was issue-bound on DaViT stages 0-1).  Same loads, same summation order.
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const T* x = (const T*)a.x + row * a.C;
  const T* ad = a.add ? (const T*)a.add + (row % a.period) * a.C : nullptr;
  float v[NIT];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    int c = lane + i * 64;
    float t = 0.0f;
    if (c < a.C) { t = ldf(x + c); if (ad) t += ldf(ad + c); }
    v[i] = t; s += t;
  }
  float mean = wave_sum(s) / (float)a.C;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    int c = lane + i * 64;
    if (c < a.C) { float d = v[i] - mean; q += d * d; }
  }
  float rstd = 1.0f / sqrtf(wave_sum(q) / (float)a.C + a.eps);
  T* y = (T*)a.y + row * a.C;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    int c = lane + i * 64;
    if (c < a.C) stf(y + c, (v[i] - mean) * rstd * a.g[c] + a.b[c]);
  }
}

template <typename T>
void launch_layernorm_typed(const LnArgs& a, unsigned blocks, hipStream_t s) {
  const int nit = (a.C + 63) / 64;
  if (nit <= 2) hipLaunchKernelGGL((layernorm_kernel<T, 2>), dim3(blocks), dim3(256), 0, s, a);
  else if (nit <= 4) hipLaunchKernelGGL((layernorm_kernel<T, 4>), dim3(blocks), dim3(256), 0, s, a);
  else if (nit <= 8) hipLaunchKernelGGL((layernorm_kernel<T, 8>), dim3(blocks), dim3(256), 0, s, a);
  else if (nit <= 12) hipLaunchKernelGGL((layernorm_kernel<T, 12>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((layernorm_kernel<T, 16>), dim3(blocks), dim3(256), 0, s, a);
}


// f32 rows with C % 4 == 0 (every LayerNorm of the captioner): 16-byte vectors per lane, LPR lanes per row (a wave
// covers two C = 128 rows at once), same two-pass statistics.  OMODE 0: f32 output; 1: "format B" split output (the
// consumer is a gemm_dma GEMM: hi/lo f16 pairs, omni_internal.h); 2: both (BART post-LN rows are also the residual).
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int NIT, int LPR, int OMODE>
__global__ __launch_bounds__(256) void layernorm_f32v4_kernel(LnArgs a) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, lr = lane % LPR, sub = lane / LPR;
  long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
  const bool live = row < a.rows;
  if (!live) row = a.rows - 1;                    // the lane stays for the shuffles and stores nothing
  const float* __restrict__ x = (const float*)a.x + row * a.C;
  const float* __restrict__ ad = a.add ? (const float*)a.add + (row % a.period) * a.C : nullptr;
  f32x4 v[NIT];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = (lr + i * LPR) * 4;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (c < a.C) {
      t = *reinterpret_cast<const f32x4*>(x + c);
      if (ad) t += *reinterpret_cast<const f32x4*>(ad + c);
    }
    v[i] = t;
    s += (t[0] + t[1]) + (t[2] + t[3]);
  }
  const float mean = group_sum<LPR>(s) / (float)a.C;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = (lr + i * LPR) * 4;
    if (c < a.C) {
      f32x4 d = v[i] - mean;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  const float rstd = 1.0f / sqrtf(group_sum<LPR>(q) / (float)a.C + a.eps);
  if (!live) return;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = (lr + i * LPR) * 4;
    if (c < a.C) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(a.g + c), b = *reinterpret_cast<const f32x4*>(a.b + c);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      if constexpr (OMODE != 1) *reinterpret_cast<f32x4*>((float*)a.y + row * a.C + c) = f32x4{o[0], o[1], o[2], o[3]};
      if constexpr (OMODE != 0) {
        uint2 hi, lo;
        omni_split4(o, hi, lo);
        unsigned char* p = (unsigned char*)(OMODE == 1 ? a.y : a.y2) + row * a.C * 4 + omni_split_off(c);
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + 32) = lo;
      }
    }
  }
}

template <int NIT, int LPR>
void launch_ln_v4_mode(const LnArgs& a, hipStream_t s) {
  const unsigned blocks = (unsigned)((a.rows + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)));
  if (a.omode == 0) hipLaunchKernelGGL((layernorm_f32v4_kernel<NIT, LPR, 0>), dim3(blocks), dim3(256), 0, s, a);
  else if (a.omode == 1) hipLaunchKernelGGL((layernorm_f32v4_kernel<NIT, LPR, 1>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((layernorm_f32v4_kernel<NIT, LPR, 2>), dim3(blocks), dim3(256), 0, s, a);
}

inline bool launch_layernorm_v4(const LnArgs& a, hipStream_t s) {
  if (a.C % 4) return false;
  if (a.C <= 128) launch_ln_v4_mode<1, 32>(a, s);
  else if (a.C <= 256) launch_ln_v4_mode<1, 64>(a, s);
  else if (a.C <= 512) launch_ln_v4_mode<2, 64>(a, s);
  else if (a.C <= 768) launch_ln_v4_mode<3, 64>(a, s);
  else launch_ln_v4_mode<4, 64>(a, s);
  return true;
}

// ------------------------------------------------------------------------------------ dwconv3 + layernorm fused
// x1 = x + depthwise3x3(x) + bias ; h = LayerNorm(x1).  One wave per pixel (C <= 1024): the conv result stays in
// registers for the statistics, so x1 is written once and never re-read (saves one full tensor read per DaViT
// half-block compared with dwconv3_kernel followed by layernorm_kernel).
struct DwLnArgs { const void* x; const void* w; const float* bias; const float* g; const float* b; void* y1; void* h;
                  int B, H, W, C; long long pixels; float eps; int osplit; };

template <typename T>
__global__ __launch_bounds__(256) void dwconv3_ln_kernel(DwLnArgs a) {
  constexpr int V = ElemTraits<T>::kVec;
  constexpr int MAXI = 1024 / (64 * V);          // channel vectors per lane (4 for f32, 2 for f16)
  struct Vec { T v[V]; };
  const int lane = threadIdx.x & 63;
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= a.pixels) return;
  const int w = (int)(pix % a.W);
  const long long t = pix / a.W;
  const int hh = (int)(t % a.H);
  const long long b = t / a.H;
  const T* __restrict__ X = (const T*)a.x;
  const T* __restrict__ Wt = (const T*)a.w;
  float y[MAXI][V];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + i * 64) * V;
#pragma unroll
    for (int e = 0; e < V; ++e) y[i][e] = 0.0f;
    if (c < a.C) {
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.0f;
      Vec ctr;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        int hi = hh + r - 1;
        if (hi < 0 || hi >= a.H) continue;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          int wi = w + q - 1;
          if (wi < 0 || wi >= a.W) continue;
          Vec xv = __builtin_bit_cast(Vec, *reinterpret_cast<const u32x4*>(X + ((b * a.H + hi) * a.W + wi) * a.C + c));
          Vec wv = __builtin_bit_cast(Vec, *reinterpret_cast<const u32x4*>(Wt + (r * 3 + q) * a.C + c));
          if (r == 1 && q == 1) ctr = xv;
#pragma unroll
          for (int e = 0; e < V; ++e) acc[e] += ElemTraits<T>::to_f32(wv.v[e]) * ElemTraits<T>::to_f32(xv.v[e]);
        }
      }
      Vec out;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        out.v[e] = ElemTraits<T>::from_f32((acc[e] + a.bias[c + e]) + ElemTraits<T>::to_f32(ctr.v[e]));
        y[i][e] = ElemTraits<T>::to_f32(out.v[e]);      // statistics on the value as stored (matches the unfused path)
        s += y[i][e];
      }
      *reinterpret_cast<u32x4*>((T*)a.y1 + pix * a.C + c) = __builtin_bit_cast(u32x4, out);
    }
  }
  const float mean = wave_sum(s) / (float)a.C;
  float q2 = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + i * 64) * V;
    if (c < a.C) {
#pragma unroll
      for (int e = 0; e < V; ++e) { float d = y[i][e] - mean; q2 += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)a.C + a.eps);
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = (lane + i * 64) * V;
    if (c < a.C) {
      Vec out;
#pragma unroll
      for (int e = 0; e < V; ++e) out.v[e] = ElemTraits<T>::from_f32((y[i][e] - mean) * rstd * a.g[c + e] + a.b[c + e]);
      if constexpr (V == 4) {
        if (a.osplit) {                     // format B for the LDS-DMA GEMM that consumes h (f32 plans)
          float o[4] = {ElemTraits<T>::to_f32(out.v[0]), ElemTraits<T>::to_f32(out.v[1]), ElemTraits<T>::to_f32(out.v[2]), ElemTraits<T>::to_f32(out.v[3])};
          uint2 hi, lo;
          omni_split4(o, hi, lo);
          unsigned char* p = (unsigned char*)a.h + pix * a.C * 4 + omni_split_off(c);
          *reinterpret_cast<uint2*>(p) = hi;
          *reinterpret_cast<uint2*>(p + 32) = lo;
          continue;
        }
      }
      *reinterpret_cast<u32x4*>((T*)a.h + pix * a.C + c) = __builtin_bit_cast(u32x4, out);
    }
  }
}

// Strip version of the fusion (f32 plans, C = 128 / 256 / 512 — DaViT stages 0-2, 97 % of the bytes of this op): the kernel above
// re-reads the 3x3 neighbourhood of every pixel through L2 (one wave per pixel, nine row loads per output row), which is why it
// lost to the two separate kernels in round 2 (142 ms vs 101 ms per step).  Here a thread owns NV 16-byte channel vectors of ONE
// image column and slides down a strip of SR output rows with a rolling 3-row window in registers: every input row is loaded once
// per strip (+2 halo rows per strip), the three horizontal taps are three loads of which two hit L1 (the neighbours' vectors),
// the conv result never leaves registers before the LayerNorm statistics (LPP lanes per pixel: shuffles only), and both outputs
// (x1 in f32 for the residual path, LN(x1) in format B or f32 for the GEMM) are written once.  HBM traffic: (1 + 2/SR) reads +
// 2 writes per element instead of ~2.5 + 1 (dwconv3_strip_kernel with DW_ROWS = 4, measured) + 1 + 1 (layernorm).
// Blocks are numbered so that horizontally adjacent blocks of a strip run on the SAME XCD (bid & 7 = XCD): their shared halo
// columns then hit that XCD's L2.  Per-element arithmetic (tap order, (acc + bias) + centre, two-pass statistics) is the unfused
// kernels': y1 is bit-identical to dwconv3_kernel, h equals layernorm_f32v4_kernel up to the summation order of the statistics.
// Measured and not kept (round 5, two one-minute A/B sessions on the MI355X, per-op profile of a 128-crop plan): a second row of loads in
// flight ahead of the arithmetic (244 instead of 214 registers at C = 512, same 2 waves per SIMD) — 15.6 vs 15.1 ms per 36 launches, no
// gain: the kernel is not bound by load latency; non-temporal stores for both outputs — 14.98 vs 14.90 ms, within the run-to-run noise.
template <int NV, int LPP, bool OSPLIT>
__global__ __launch_bounds__(256) void dwln_strip_kernel(DwLnArgs a, int SR, int pxb, int strips, unsigned nblocks) {
  constexpr int PPB = 256 / LPP;                                  // pixels (columns) per block
  const unsigned per = (gridDim.x + 7u) >> 3;
  const unsigned L = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);   // logical block: contiguous ranges per XCD
  if (L >= nblocks) return;
  const int pw = (int)(L % (unsigned)pxb);
  const unsigned t = L / (unsigned)pxb;
  const int strip = (int)(t % (unsigned)strips);
  const int b = (int)(t / (unsigned)strips);
  const int tid = threadIdx.x;
  const int w = pw * PPB + tid / LPP, l = tid % LPP;
  if (w >= a.W) return;                                           // whole LPP-lane groups leave together
  const int h0 = strip * SR, h1 = min(h0 + SR, a.H);
  const float* __restrict__ X = (const float*)a.x + (long long)b * a.H * a.W * a.C;
  float* __restrict__ Y1 = (float*)a.y1 + (long long)b * a.H * a.W * a.C;
  unsigned char* __restrict__ Hh = (unsigned char*)a.h + (long long)b * a.H * a.W * a.C * 4;
  const float* __restrict__ Wt = (const float*)a.w;
  f32x4 wt[9][NV], cb[NV], lg[NV], lb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (l + i * LPP) * 4;
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k][i] = *reinterpret_cast<const f32x4*>(Wt + k * a.C + c);
    cb[i] = *reinterpret_cast<const f32x4*>(a.bias + c);
    lg[i] = *reinterpret_cast<const f32x4*>(a.g + c);
    lb[i] = *reinterpret_cast<const f32x4*>(a.b + c);
  }
  const bool wl = w > 0, wr = w + 1 < a.W;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  auto load_row = [&](int h, f32x4 (&dst)[3][NV]) {
    const bool hok = h >= 0 && h < a.H;
    const float* __restrict__ row = X + ((long long)(hok ? h : 0) * a.W + w) * a.C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (l + i * LPP) * 4;
      dst[0][i] = (hok && wl) ? *reinterpret_cast<const f32x4*>(row - a.C + c) : z4;
      dst[1][i] = hok ? *reinterpret_cast<const f32x4*>(row + c) : z4;
      dst[2][i] = (hok && wr) ? *reinterpret_cast<const f32x4*>(row + a.C + c) : z4;
    }
  };
  f32x4 r0[3][NV], r1[3][NV], r2[3][NV], nx[3][NV];
  load_row(h0 - 1, r0);
  load_row(h0, r1);
  load_row(h0 + 1, r2);
  const float invC = 1.0f / (float)a.C;
  for (int h = h0; h < h1; ++h) {
    if (h + 1 < h1) load_row(h + 2, nx);                          // in flight under this row's arithmetic and stores
    f32x4 y[NV];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 acc = z4;
#pragma unroll
      for (int q = 0; q < 3; ++q) acc += wt[0 + q][i] * r0[q][i];
#pragma unroll
      for (int q = 0; q < 3; ++q) acc += wt[3 + q][i] * r1[q][i];
#pragma unroll
      for (int q = 0; q < 3; ++q) acc += wt[6 + q][i] * r2[q][i];
      y[i] = (acc + cb[i]) + r1[1][i];
      s += (y[i][0] + y[i][1]) + (y[i][2] + y[i][3]);
    }
    const long long pix = (long long)h * a.W + w;
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(Y1 + pix * a.C + (l + i * LPP) * 4) = y[i];
    const float mean = group_sum<LPP>(s) * invC;
    float q2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const f32x4 d = y[i] - mean;
      q2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    const float rstd = 1.0f / sqrtf(group_sum<LPP>(q2) * invC + a.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (l + i * LPP) * 4;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (y[i][e] - mean) * rstd * lg[i][e] + lb[i][e];
      if constexpr (OSPLIT) {
        uint2 hi, lo;
        omni_split4(o, hi, lo);
        unsigned char* p = Hh + pix * a.C * 4 + omni_split_off(c);
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + 32) = lo;
      } else {
        *reinterpret_cast<f32x4*>(Hh + (pix * a.C + c) * 4) = f32x4{o[0], o[1], o[2], o[3]};
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < NV; ++i) { r0[q][i] = r1[q][i]; r1[q][i] = r2[q][i]; r2[q][i] = nx[q][i]; }
  }
}

template <int NV, int LPP>
void launch_dwln_strip(const DwLnArgs& a, hipStream_t s) {
  constexpr int PPB = 256 / LPP;
  const int pxb = (a.W + PPB - 1) / PPB;
  // strip length: long strips re-read fewer halo rows (2 / SR); short ones keep >= ~1024 blocks in flight on small batches
  int SR = 16;
  while (SR > 4 && (long long)a.B * ((a.H + SR - 1) / SR) * pxb < 1024) SR >>= 1;
  const int strips = (a.H + SR - 1) / SR;
  const long long nb = (long long)a.B * strips * pxb;
  const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);
  if (a.osplit) hipLaunchKernelGGL((dwln_strip_kernel<NV, LPP, true>), dim3(grid), dim3(256), 0, s, a, SR, pxb, strips, (unsigned)nb);
  else hipLaunchKernelGGL((dwln_strip_kernel<NV, LPP, false>), dim3(grid), dim3(256), 0, s, a, SR, pxb, strips, (unsigned)nb);
}

// ------------------------------------------------------------------------------------ attn_rows
struct AttnArgs {
  const void* q; const void* k; const void* v; void* o; const float* kbias; const float* vbias;
  int ldq, ldk, ldv, ldo, qoff, koff, voff, ooff;   // element strides / channel offsets
  int heads, nq, nk, groups;                         // per group: nq queries, nk keys
  int mode, H, W, wy, wx;                            // window mode: image H x W, wy x wx windows of 12x12
  float scale;
  int osplit;                                        // f32 plans: write o in "format B" (hi|lo f16 pairs) for the LDS-DMA GEMM that follows
};

// store one f32 output element of channel c of row `row` (f32 elements per row: ld): plain, or as the two halves of format B
template <typename T>
__device__ __forceinline__ void store_out(void* base, long long row, int ld, int c, float v, int osplit) {
  if (osplit) {
    unsigned short hi, lo;
    omni_split1(v, hi, lo);
    unsigned short* p = (unsigned short*)base + row * ld * 2 + omni_split_half_index(c);
    p[0] = hi;
    p[16] = lo;
  } else {
    stf((T*)base + row * ld + c, v);
  }
}

// token row (into the [B*H*W] token matrix) of window-local index i (0..143), or -1 for padding
__device__ __forceinline__ long long window_row(const AttnArgs& a, int g, int i) {
  int wpi = a.wy * a.wx;
  int b = g / wpi, wrem = g - b * wpi;
  int wyi = wrem / a.wx, wxi = wrem - wyi * a.wx;
  int r = wyi * 12 + i / 12, c = wxi * 12 + i % 12;
  if (r >= a.H || c >= a.W) return -1;
  return ((long long)b * a.H + r) * a.W + c;
}

template <typename T, int D, int NT>
__global__ __launch_bounds__(NT) void attn_rows_kernel(AttnArgs a) {
  constexpr int KT = 48;                       // keys per LDS tile (144 = 3 tiles)
  __shared__ __attribute__((aligned(16))) float sk[KT][D];
  __shared__ __attribute__((aligned(16))) float sv[KT][D];
  const int g = blockIdx.z, h = blockIdx.y;
  const int qi = blockIdx.x * NT + threadIdx.x;
  const T* Q = (const T*)a.q; const T* K = (const T*)a.k; const T* V = (const T*)a.v;
  long long qrow = -1;
  if (qi < a.nq) qrow = a.mode == 1 ? window_row(a, g, qi) : (long long)g * a.nq + qi;
  float q[D], o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { q[d] = qrow >= 0 ? ldf(Q + qrow * a.ldq + a.qoff + h * D + d) : 0.0f; o[d] = 0.0f; }
  float m = -INFINITY, l = 0.0f;
  for (int k0 = 0; k0 < a.nk; k0 += KT) {
    __syncthreads();
    for (int e = threadIdx.x; e < KT * (D / 4); e += NT) {     // one 4-float group per lane
      int kk = e / (D / 4), d = (e - kk * (D / 4)) * 4;
      int ki = k0 + kk;
      float kq[4] = {0.f, 0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f};
      if (ki < a.nk) {
        long long krow = a.mode == 1 ? window_row(a, g, ki) : (long long)g * a.nk + ki;
        if (krow >= 0) {
          const T* kp = K + krow * a.ldk + a.koff + h * D + d;
          const T* vp = V + krow * a.ldv + a.voff + h * D + d;
#pragma unroll
          for (int u = 0; u < 4; ++u) { kq[u] = ldf(kp + u); vq[u] = ldf(vp + u); }
        } else {            // zero-padded window token: qkv(0) = bias, NOT masked (hf :345,372-379)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            kq[u] = a.kbias ? a.kbias[h * D + d + u] : 0.0f;
            vq[u] = a.vbias ? a.vbias[h * D + d + u] : 0.0f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { sk[kk][d + u] = kq[u]; sv[kk][d + u] = vq[u]; }
    }
    __syncthreads();
    int lim = a.nk - k0 < KT ? a.nk - k0 : KT;
    // online softmax in chunks of 8 keys: one running-max update / accumulator rescale per chunk
    for (int kk0 = 0; kk0 < lim; kk0 += 8) {
      float sc[8];
      float cm = -INFINITY;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float s = -INFINITY;
        if (kk0 + u < lim) {
          s = 0.0f;
#pragma unroll
          for (int d = 0; d < D; ++d) s = fmaf(q[d], sk[kk0 + u][d], s);
          s *= a.scale;
        }
        sc[u] = s;
        cm = fmaxf(cm, s);
      }
      float mn = fmaxf(m, cm);
      float corr = __expf(m - mn);         // m = -inf on the first chunk -> 0
      l *= corr;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] *= corr;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (kk0 + u < lim) {
          float p = __expf(sc[u] - mn);
          l += p;
#pragma unroll
          for (int d = 0; d < D; ++d) o[d] = fmaf(p, sv[kk0 + u][d], o[d]);
        }
      }
      m = mn;
    }
  }
  if (qrow >= 0) {
    T* O = (T*)a.o + qrow * a.ldo + a.ooff + h * D;
    float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) stf(O + d, o[d] * inv);
  }
}

// ------------------------------------------------------------------------------------ window attention on MFMA
// DaViT 12x12 window attention (hf:models/florence2/modeling_florence2.py:338-398) with both contractions on the
// matrix cores at f32-class accuracy (split-f16 operands, see conv_igemm.hip):
//   S^T = K Q^T  via v_mfma_f32_16x16x32_f16 (head_dim 32 = one K step): each lane ends up with ONE query
//         (column = lane&15) and 4 keys per 16-key tile (rows 4*(lane>>4)+reg), so the softmax row reduction is
//         in-register plus two xor-shuffles (16, 32) — no LDS round trip for the scores;
//   O   = P V    the lane's exponentials ARE the A-operand fragment of P (8 keys per 32-key block in a fixed
//         permuted order); V^T is staged in LDS so the matching B fragment is two 8-byte reads.
// One workgroup (3 waves) per (window, head): K and V^T (hi|lo f16) live in LDS, Q fragments go straight from
// HBM to registers; wave w owns query tiles w, w+3, w+6 of the 9 x 16 queries.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split1(float a, half_t& h, half_t& l) {
  h = (half_t)a;
  l = (half_t)((a - (float)h) * 2048.0f);
}

template <typename T>
__global__ __launch_bounds__(192) void window_attn_mfma_kernel(AttnArgs a) {
  constexpr int D = 32, NKP = 160;                    // 144 keys padded to 10 tiles of 16
  constexpr int KROW = 80;                            // bytes per K row (32 halves + 16 pad)
  constexpr int VROW = 336;                           // bytes per V^T row (160 halves + 16 pad)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * NKP * KROW + 2 * D * VROW];
  unsigned char* Kh = lds;
  unsigned char* Kl = lds + NKP * KROW;
  unsigned char* Vh = lds + 2 * NKP * KROW;
  unsigned char* Vl = Vh + D * VROW;
  const int g = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* Qp = (const T*)a.q; const T* Kp = (const T*)a.k; const T* Vp = (const T*)a.v;

  // ---- stage K (row-major) and V^T, split into hi|lo halves
  for (int e = tid; e < NKP * (D / 4); e += 192) {
    int key = e / (D / 4), d0 = (e - key * (D / 4)) * 4;
    float kv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
    if (key < 144) {
      long long row = window_row(a, g, key);
      if (row >= 0) {
        if constexpr (sizeof(T) == 4) {      // 16-byte loads (qkv rows are 16-byte aligned: C % 4 == 0)
          f32x4 k4 = *reinterpret_cast<const f32x4*>(Kp + row * a.ldk + a.koff + h * D + d0);
          f32x4 v4 = *reinterpret_cast<const f32x4*>(Vp + row * a.ldv + a.voff + h * D + d0);
#pragma unroll
          for (int u = 0; u < 4; ++u) { kv[u] = k4[u]; vv[u] = v4[u]; }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            kv[u] = ldf(Kp + row * a.ldk + a.koff + h * D + d0 + u);
            vv[u] = ldf(Vp + row * a.ldv + a.voff + h * D + d0 + u);
          }
        }
      } else {                               // zero-padded window token: qkv(0) = bias, not masked
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          kv[u] = a.kbias ? a.kbias[h * D + d0 + u] : 0.0f;
          vv[u] = a.vbias ? a.vbias[h * D + d0 + u] : 0.0f;
        }
      }
    }
    half_t kh[4], kl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      split1(kv[u], kh[u], kl[u]);
      half_t vh, vl;
      split1(vv[u], vh, vl);
      *reinterpret_cast<half_t*>(Vh + (d0 + u) * VROW + key * 2) = vh;
      *reinterpret_cast<half_t*>(Vl + (d0 + u) * VROW + key * 2) = vl;
    }
    h16x4 k4h = {kh[0], kh[1], kh[2], kh[3]}, k4l = {kl[0], kl[1], kl[2], kl[3]};
    *reinterpret_cast<h16x4*>(Kh + key * KROW + d0 * 2) = k4h;
    *reinterpret_cast<h16x4*>(Kl + key * KROW + d0 * 2) = k4l;
  }
  __syncthreads();

  const int qc = lane & 15, grp = lane >> 4;
  const float inv2048 = 1.0f / 2048.0f;
  for (int qt = wave; qt < 9; qt += 3) {
    // ---- Q fragment of this lane: query qt*16+qc, d = 8*grp .. 8*grp+7
    const int qi = qt * 16 + qc;
    const long long qrow = window_row(a, g, qi);
    h16x8 qh, ql;
    {
      float qv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (qrow >= 0) {
        const T* qp = Qp + qrow * a.ldq + a.qoff + h * D + grp * 8;
        if constexpr (sizeof(T) == 4) {
          f32x4 q0 = *reinterpret_cast<const f32x4*>(qp), q1 = *reinterpret_cast<const f32x4*>(qp + 4);
#pragma unroll
          for (int u = 0; u < 4; ++u) { qv[u] = q0[u]; qv[4 + u] = q1[u]; }
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) qv[u] = ldf(qp + u);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        half_t hh, ll;
        split1(qv[u], hh, ll);
        qh[u] = hh; ql[u] = ll;
      }
    }
    // ---- S^T tiles: 10 key tiles x (1 + 2) MFMAs
    float sc[40];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 10; ++kt) {
      const unsigned char* kr = Kh + (kt * 16 + qc) * KROW + grp * 16;
      h16x8 kh = *reinterpret_cast<const h16x8*>(kr);
      h16x8 kl = *reinterpret_cast<const h16x8*>(kr + NKP * KROW);
      f32x4 accM = {0.f, 0.f, 0.f, 0.f}, accC = {0.f, 0.f, 0.f, 0.f};
      accM = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh, accM, 0, 0, 0);
      accC = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql, accC, 0, 0, 0);
      accC = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh, accC, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = kt * 16 + grp * 4 + r;
        float sv = (accM[r] + accC[r] * inv2048) * a.scale;
        sv = key < 144 ? sv : -INFINITY;
        sc[kt * 4 + r] = sv;
        mx = fmaxf(mx, sv);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < 40; ++e) { sc[e] = __expf(sc[e] - mx); sum += sc[e]; }   // v_exp_f32: rel. error ~|x|*6e-8
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    // ---- O = P V : 5 key blocks of 32 x 2 d-tiles
    f32x4 oM[2], oC[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { oM[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; oC[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      h16x8 ph, pl;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        half_t hh, ll;
        split1(sc[(2 * j + (u >> 2)) * 4 + (u & 3)], hh, ll);
        ph[u] = hh; pl[u] = ll;
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const unsigned char* vr = Vh + (dt * 16 + qc) * VROW + (32 * j + 4 * grp) * 2;
        h16x4 a0 = *reinterpret_cast<const h16x4*>(vr), a1 = *reinterpret_cast<const h16x4*>(vr + 32);
        h16x4 b0 = *reinterpret_cast<const h16x4*>(vr + D * VROW), b1 = *reinterpret_cast<const h16x4*>(vr + D * VROW + 32);
        h16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        h16x8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        oM[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vh, oM[dt], 0, 0, 0);
        oC[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vl, oC[dt], 0, 0, 0);
        oC[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl, vh, oC[dt], 0, 0, 0);
      }
    }
    // ---- normalise and store: this lane holds O[query 4*grp+r][d = dt*16+qc]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int ql_ = grp * 4 + r;
      float rs = __shfl(sum, ql_);                         // row sum of query ql_ (replicated in every lane group)
      long long orow = window_row(a, g, qt * 16 + ql_);
      if (orow >= 0) {
        float inv = 1.0f / rs;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          store_out<T>(a.o, orow, a.ldo, a.ooff + h * D + dt * 16 + qc, (oM[dt][r] + oC[dt][r] * inv2048) * inv, a.osplit);
      }
    }
  }
}

// Pair-wise f32 -> (hi, lo) splits of the f32 attention kernels below (v_cvt_pkrtz + packed f32 sub / mul: 3 instructions per value
// instead of 5 for the one-value-at-a-time form of the generic kernel above).  hi halves are round-toward-zero; the lo half (scaled by
// 2048) absorbs the remainder either way.
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& l) {
  typedef __fp16 hv2 __attribute__((ext_vector_type(2)));
  const hv2 hh = __builtin_amdgcn_cvt_pkrtz(a, b);
  const f32x2 r = (f32x2{a, b} - f32x2{(float)hh[0], (float)hh[1]}) * 2048.0f;
  const hv2 ll = __builtin_amdgcn_cvt_pkrtz(r[0], r[1]);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}
__device__ __forceinline__ void split4v(const f32x4& v, uint2& h, uint2& l) {
  split2(v[0], v[1], h.x, l.x);
  split2(v[2], v[3], h.y, l.y);
}

// split2 with the remainder as one mixed-precision fma per value (v_fma_mix_f32 reads the f16 hi half directly: no v_cvt_f32_f16, no
// subtraction): 2048 (x - hi) = fma(hi, -2048, 2048 x), exact like the form above (2048 x is exact, x - hi has <= 13 significant bits)
__device__ __forceinline__ void split2m(float a, float b, unsigned& h, unsigned& l) {
  typedef __fp16 hv2 __attribute__((ext_vector_type(2)));
  const hv2 hh = __builtin_amdgcn_cvt_pkrtz(a, b);
  const f32x2 s = f32x2{a, b} * 2048.0f;
  const hv2 ll = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)hh[0], -2048.0f, s[0]), __builtin_fmaf((float)hh[1], -2048.0f, s[1]));
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}
__device__ __forceinline__ void split4m(const f32x4& v, uint2& h, uint2& l) {
  split2m(v[0], v[1], h.x, l.x);
  split2m(v[2], v[3], h.y, l.y);
}

// The f32 window kernel (adopted in round 4: 2.45 vs 3.69 ms per launch at 4.7 M tokens, 3.7-3.9 TB/s instead of 2.4-2.6;
// profiles/r4_s2_per_op_*.txt).  Its round-3 predecessor issued 2 319 VALU instructions
// per wave for 171 MFMAs (profiles/r3_static_isa_report.txt): it is bound by its vector ALU, not by HBM (2.1-2.3 TB/s) or the matrix
// pipe.  Same LDS image, same split-f16 x3 arithmetic, same blocks; what changes is where the VALU instructions went:
//   * token addressing: ~30 calls of row_of() per thread (two divisions by 12, a 64-bit row index, a 64-bit multiply by the row
//     pitch each).  Here the window origin is ONE uniform 64-bit pointer per tensor and every token of the window is a 32-bit element
//     offset from it; the divisions by 12 are done once per thread (a thread's K items are 24 keys = 2 window rows apart, its V
//     items 8 rows, its Q tiles 4 rows);
//   * P V: the MFMA takes V^T as the ROW operand and P as the column operand (the per-lane register contents are the same, only the
//     roles swap), so the accumulator holds O^T — each lane owns ONE query and 4 consecutive channels: no __shfl of the row sum, one
//     output row address per tile, 16-byte f32 stores / 8-byte hi and lo stores instead of sixteen 2- or 4-byte stores per tile;
//   * scores: the 10th key tile (keys 144..159) is padding for every window — not computed; log2(e) is folded into the Q scale so an
//     exponential is v_sub + v_exp; merges, maxima and sums are written on pairs (packed f32 instructions).
// Results agree with the kernel above to f32 rounding (the exponentials see scores scaled before the split instead of after).
__global__ __launch_bounds__(192, 2) void window_attn_mfma_f32_kernel(AttnArgs a) {
  constexpr int D = 32, NKP = 160, KROW = 80, VROW = 336;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * NKP * KROW + 2 * D * VROW];
  unsigned char* Kh = lds;
  unsigned char* Kl = lds + NKP * KROW;
  unsigned char* Vh = lds + 2 * NKP * KROW;
  unsigned char* Vl = Vh + D * VROW;
  const int g = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // window origin (uniform).  Token (wr, wc) of the window, 0 <= wr, wc < 12, lies wr * W + wc rows behind it and exists iff
  // wr < rows_left and wc < cols_left (windows of the last row / column of a padded image are cut)
  const int wpi = a.wy * a.wx;
  const int b = g / wpi, wrem = g - b * wpi;
  const int wyi = wrem / a.wx, wxi = wrem - wyi * a.wx;
  const int r0 = wyi * 12, c0 = wxi * 12;
  const int rows_left = a.H - r0, cols_left = a.W - c0;
  const long long row00 = ((long long)b * a.H + r0) * a.W + c0;
  const float* Qb = (const float*)a.q + row00 * a.ldq + a.qoff + h * D;
  const float* Kb = (const float*)a.k + row00 * a.ldk + a.koff + h * D;
  const float* Vb = (const float*)a.v + row00 * a.ldv + a.voff + h * D;
  const int qc = lane & 15, grp = lane >> 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // ---- 1. all global loads of the block
  // Q: tile t of this wave holds window tokens 16 * (wave + 3 t) + qc = iq0 + 48 t: 4 t window rows below token iq0
  const int iq0 = 16 * wave + qc;
  const int qwr = iq0 / 12, qwc = iq0 - 12 * qwr;
  const int qtok = qwr * a.W + qwc;
  f32x4 qraw[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const bool ok = qwr + 4 * t < rows_left && qwc < cols_left;
    const float* qp = Qb + (ok ? (qtok + 4 * t * a.W) * a.ldq : 0) + grp * 8;
    qraw[t][0] = ok ? *reinterpret_cast<const f32x4*>(qp) : z4;
    qraw[t][1] = ok ? *reinterpret_cast<const f32x4*>(qp + 4) : z4;
  }
  // K: item e = tid + 192 j = (key, 4-channel group) with key = t8 + 24 j: 2 j window rows below key t8
  const int t8 = tid >> 3, d0 = (tid & 7) * 4;
  const int kwr = t8 >= 12 ? 1 : 0, kwc = t8 - 12 * kwr;
  const int ktok = kwr * a.W + kwc;
  f32x4 kpad = z4, vpad = z4;
  if (a.kbias) kpad = *reinterpret_cast<const f32x4*>(a.kbias + h * D + d0);
  if (a.vbias) vpad = *reinterpret_cast<const f32x4*>(a.vbias + h * D + d0);
  f32x4 kraw[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const bool ok = kwr + 2 * j < rows_left && kwc < cols_left;
    kraw[j] = ok ? *reinterpret_cast<const f32x4*>(Kb + (ktok + 2 * j * a.W) * a.ldk + d0) : kpad;
  }
  // V: item e = tid + 192 j = (key quad kq = t8 + 24 j, 4-channel group): keys 4 kq + u = window row t8 / 3 + 8 j, columns 4 (t8 % 3) + u
  const int vwr = t8 / 3, vwc = 4 * (t8 - 3 * vwr);
  const int vtok = vwr * a.W + vwc;
  f32x4 vraw[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (j == 0 || tid < 96) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = vwr + 8 * j < rows_left && vwc + u < cols_left;
        vraw[j][u] = ok ? *reinterpret_cast<const f32x4*>(Vb + (vtok + 8 * j * a.W + u) * a.ldv + d0) : vpad;
      }
    }
  }
  // ---- 2. LDS image: K rows (hi | lo), V^T rows (hi | lo), keys 144..159 zero (identical to the kernel above)
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int key = t8 + 24 * j;
    uint2 kh, kl;
    split4m(kraw[j], kh, kl);
    *reinterpret_cast<uint2*>(Kh + key * KROW + d0 * 2) = kh;
    *reinterpret_cast<uint2*>(Kl + key * KROW + d0 * 2) = kl;
  }
  if (tid >= 128) {                                   // V^T columns 144..159 of all 32 rows (hi and lo): 64 threads x 16 halves
    const int r = tid - 128;                          // (the K rows 144..159 are never read: the 10th score tile is not computed)
    unsigned char* p = (r < 32 ? Vh : Vl) + (r & 31) * VROW + 144 * 2;
    const u32x4 z = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(p) = z;
    *reinterpret_cast<u32x4*>(p + 16) = z;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int kq = t8 + 24 * j;
    if (j == 0 || tid < 96) {
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {                // channel d0 + dd: its four keys 4kq .. 4kq+3
        uint2 vh, vl;
        split4m(f32x4{vraw[j][0][dd], vraw[j][1][dd], vraw[j][2][dd], vraw[j][3][dd]}, vh, vl);
        *reinterpret_cast<uint2*>(Vh + (d0 + dd) * VROW + kq * 8) = vh;
        *reinterpret_cast<uint2*>(Vl + (d0 + dd) * VROW + kq * 8) = vl;
      }
    }
  }
  __syncthreads();

  const float inv2048 = 1.0f / 2048.0f;
  const float qscale = a.scale * 1.4426950408889634f;  // scores in units of log2: exp(s - m) = exp2(s' - m')
  unsigned char* const Ob = (unsigned char*)a.o + (row00 * a.ldo + a.ooff + h * D) * 4;   // a split row has the f32 row's pitch
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    u32x4 qhu, qlu;                                   // Q fragment (8 halves each), pre-multiplied by the softmax scale
    {
      const f32x4 q0 = qraw[t][0] * qscale, q1 = qraw[t][1] * qscale;
      uint2 h0, l0, h1, l1;
      split4m(q0, h0, l0);
      split4m(q1, h1, l1);
      qhu = u32x4{h0.x, h0.y, h1.x, h1.y};
      qlu = u32x4{l0.x, l0.y, l1.x, l1.y};
    }
    const h16x8 qh = __builtin_bit_cast(h16x8, qhu), ql = __builtin_bit_cast(h16x8, qlu);
    f32x2 sc[18];                                     // scores of query qc against keys 16 kt + 4 grp + {0..3}, kt = 0..8
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 9; ++kt) {
      const unsigned char* kr = Kh + (kt * 16 + qc) * KROW + grp * 16;
      const h16x8 kh = *reinterpret_cast<const h16x8*>(kr);
      const h16x8 kl = *reinterpret_cast<const h16x8*>(kr + NKP * KROW);
      f32x4 accM = z4, accC = z4;
      accM = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh, accM, 0, 0, 0);
      accC = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql, accC, 0, 0, 0);
      accC = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh, accC, 0, 0, 0);
      sc[2 * kt] = f32x2{accC[0], accC[1]} * inv2048 + f32x2{accM[0], accM[1]};
      sc[2 * kt + 1] = f32x2{accC[2], accC[3]} * inv2048 + f32x2{accM[2], accM[3]};
      mx = fmaxf(fmaxf(mx, sc[2 * kt][0]), sc[2 * kt][1]);           // v_max3_f32
      mx = fmaxf(fmaxf(mx, sc[2 * kt + 1][0]), sc[2 * kt + 1][1]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 18; ++e) {
      const f32x2 d = sc[e] - mx;
      sc[e] = f32x2{__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
      sum2 += sc[e];
    }
    float sum = sum2[0] + sum2[1];
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);                       // row sum of query qc, in every lane that holds a part of it
    f32x4 oM[2], oC[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { oM[dt] = z4; oC[dt] = z4; }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      unsigned h0, l0, h1, l1, h2 = 0u, l2 = 0u, h3 = 0u, l3 = 0u;
      split2m(sc[4 * j][0], sc[4 * j][1], h0, l0);
      split2m(sc[4 * j + 1][0], sc[4 * j + 1][1], h1, l1);
      if (j < 4) {                                    // j = 4: slots 4..7 are keys 144..159 (padding): P = 0
        split2m(sc[4 * j + 2][0], sc[4 * j + 2][1], h2, l2);
        split2m(sc[4 * j + 3][0], sc[4 * j + 3][1], h3, l3);
      }
      const h16x8 ph = __builtin_bit_cast(h16x8, u32x4{h0, h1, h2, h3});
      const h16x8 pl = __builtin_bit_cast(h16x8, u32x4{l0, l1, l2, l3});
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const unsigned char* vr = Vh + (dt * 16 + qc) * VROW + (32 * j + 4 * grp) * 2;
        const uint2 a0 = *reinterpret_cast<const uint2*>(vr), a1 = *reinterpret_cast<const uint2*>(vr + 32);
        const uint2 b0 = *reinterpret_cast<const uint2*>(vr + D * VROW), b1 = *reinterpret_cast<const uint2*>(vr + D * VROW + 32);
        const h16x8 vh = __builtin_bit_cast(h16x8, u32x4{a0.x, a0.y, a1.x, a1.y});
        const h16x8 vl = __builtin_bit_cast(h16x8, u32x4{b0.x, b0.y, b1.x, b1.y});
        // O^T[channel dt * 16 + 4 grp + r][query qc]: V^T is the row operand
        oM[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, oM[dt], 0, 0, 0);
        oC[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, oC[dt], 0, 0, 0);
        oC[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, oC[dt], 0, 0, 0);
      }
    }
    // ---- normalise and store: this lane holds channels dt * 16 + 4 grp + 0..3 of query qc
    if (qwr + 4 * t < rows_left && qwc < cols_left) {
      const float inv = 1.0f / sum;
      unsigned char* orow = Ob + (long long)(qtok + 4 * t * a.W) * a.ldo * 4;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f32x4 o = (oC[dt] * inv2048 + oM[dt]) * inv;
        const int c = dt * 16 + 4 * grp;
        if (a.osplit) {
          const float v[4] = {o[0], o[1], o[2], o[3]};
          uint2 hi, lo;
          omni_split4(v, hi, lo);
          *reinterpret_cast<uint2*>(orow + omni_split_off(c)) = hi;
          *reinterpret_cast<uint2*>(orow + omni_split_off(c) + 32) = lo;
        } else {
          *reinterpret_cast<f32x4*>(orow + c * 4) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------ plain MHA on MFMA (BART encoder)
// softmax(q k^T * scale) v for head_dim 64 and any key count (hf:models/bart/modeling_bart.py:143-257), flash-style:
// keys are walked in blocks of 32 with an online softmax; both contractions use split-f16 MFMA exactly as in
// window_attn_mfma_kernel (S^T = K Q^T so the lane owns one query; the lane's exponentials are the P fragment;
// V^T staged in LDS).  One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns two 16-query tiles.
template <typename T>
__global__ __launch_bounds__(256, 2) void mha_mfma_kernel(AttnArgs a) {
  constexpr int D = 64, KB = 32;
  constexpr int KROW = 144;                     // bytes per staged K row (64 halves + 16 pad)
  constexpr int VROW = 80;                      // bytes per staged V^T row (32 halves + 16 pad)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * KB * KROW + 2 * D * VROW];
  unsigned char* Kh = lds;
  unsigned char* Kl = lds + KB * KROW;
  unsigned char* Vh = lds + 2 * KB * KROW;
  unsigned char* Vl = Vh + D * VROW;
  const int g = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qc = lane & 15, grp = lane >> 4;
  const T* Qp = (const T*)a.q; const T* Kp = (const T*)a.k; const T* Vp = (const T*)a.v;
  const float inv2048 = 1.0f / 2048.0f;

  // Q fragments of the wave's two query tiles: query q0 + t*16 + qc, d = 32*ks + 8*grp .. +7
  const int q0 = blockIdx.x * 128 + wave * 32;
  h16x8 qh[2][2], ql[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int qi = q0 + t * 16 + qc;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float v = qi < a.nq ? ldf(Qp + ((long long)g * a.nq + qi) * a.ldq + a.qoff + h * D + ks * 32 + grp * 8 + u) : 0.0f;
        half_t hh, ll;
        split1(v, hh, ll);
        qh[t][ks][u] = hh; ql[t][ks][u] = ll;
      }
    }
  }
  f32x4 oM[2][4], oC[2][4];
  float m[2], lsum[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    m[t] = -INFINITY; lsum[t] = 0.0f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { oM[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; oC[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  }

  for (int k0 = 0; k0 < a.nk; k0 += KB) {
    __syncthreads();
    // ---- stage 32 keys: K row-major, V transposed, hi|lo halves
    for (int e = tid; e < KB * (D / 4); e += 256) {
      int key = e / (D / 4), d0 = (e - key * (D / 4)) * 4;
      float kv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
      if (k0 + key < a.nk) {
        long long row = (long long)g * a.nk + k0 + key;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          kv[u] = ldf(Kp + row * a.ldk + a.koff + h * D + d0 + u);
          vv[u] = ldf(Vp + row * a.ldv + a.voff + h * D + d0 + u);
        }
      }
      half_t kh[4], kl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        split1(kv[u], kh[u], kl[u]);
        half_t vh, vl;
        split1(vv[u], vh, vl);
        *reinterpret_cast<half_t*>(Vh + (d0 + u) * VROW + key * 2) = vh;
        *reinterpret_cast<half_t*>(Vl + (d0 + u) * VROW + key * 2) = vl;
      }
      h16x4 k4h = {kh[0], kh[1], kh[2], kh[3]}, k4l = {kl[0], kl[1], kl[2], kl[3]};
      *reinterpret_cast<h16x4*>(Kh + key * KROW + d0 * 2) = k4h;
      *reinterpret_cast<h16x4*>(Kl + key * KROW + d0 * 2) = k4l;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // ---- S^T for 2 key tiles x 2 d-steps
      float sc[8];
      float bm = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        f32x4 accM = {0.f, 0.f, 0.f, 0.f}, accC = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned char* kr = Kh + (kt * 16 + qc) * KROW + ks * 64 + grp * 16;
          h16x8 kh = *reinterpret_cast<const h16x8*>(kr);
          h16x8 kl = *reinterpret_cast<const h16x8*>(kr + KB * KROW);
          accM = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[t][ks], accM, 0, 0, 0);
          accC = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[t][ks], accC, 0, 0, 0);
          accC = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[t][ks], accC, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int key = k0 + kt * 16 + grp * 4 + r;
          float sv = (accM[r] + accC[r] * inv2048) * a.scale;
          sv = key < a.nk ? sv : -INFINITY;
          sc[kt * 4 + r] = sv;
          bm = fmaxf(bm, sv);
        }
      }
      bm = fmaxf(bm, __shfl_xor(bm, 16));
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      float mn = fmaxf(m[t], bm);
      float corr = __expf(m[t] - mn);
      m[t] = mn;
      float ps = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = __expf(sc[e] - mn); ps += sc[e]; }
      lsum[t] = lsum[t] * corr + ps;            // per-lane partial (this lane's keys); reduced at the end
      // rescale O rows: this lane holds O[query 4*grp+r][.] -> need that query's corr
      float cr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) cr[r] = __shfl(corr, grp * 4 + r);
      h16x8 ph, pl;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        half_t hh, ll;
        split1(sc[u], hh, ll);
        ph[u] = hh; pl[u] = ll;
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { oM[t][dt][r] *= cr[r]; oC[t][dt][r] *= cr[r]; }
        const unsigned char* vr = Vh + (dt * 16 + qc) * VROW + (4 * grp) * 2;
        h16x4 a0 = *reinterpret_cast<const h16x4*>(vr), a1 = *reinterpret_cast<const h16x4*>(vr + 32);
        h16x4 b0 = *reinterpret_cast<const h16x4*>(vr + D * VROW), b1 = *reinterpret_cast<const h16x4*>(vr + D * VROW + 32);
        h16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        h16x8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        oM[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vh, oM[t][dt], 0, 0, 0);
        oC[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vl, oC[t][dt], 0, 0, 0);
        oC[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl, vh, oC[t][dt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float tot = lsum[t];
    tot += __shfl_xor(tot, 16);
    tot += __shfl_xor(tot, 32);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int qi = q0 + t * 16 + grp * 4 + r;
      float rs = __shfl(tot, grp * 4 + r);
      if (qi < a.nq) {
        float inv = 1.0f / rs;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          store_out<T>(a.o, (long long)g * a.nq + qi, a.ldo, a.ooff + h * D + dt * 16 + qc, (oM[t][dt][r] + oC[t][dt][r] * inv2048) * inv, a.osplit);
      }
    }
  }
}


// f32 plans (adopted in round 4: 4.07 vs 6.84 ms for the six encoder layers of a 128-crop batch, 198 vs 118 TF/s; profiles/r4_s2_per_op_*.txt).
// The generic kernel above (f16 plans) stages 32 keys at a time
// with 4-byte loads, converts one value per instruction, writes V^T to LDS two bytes at a time and brackets every 32 keys with two
// block barriers and no load in flight across them: at 585 keys a block spends its 19 iterations mostly waiting (1.0 ms per launch
// at 134.6 GFLOP = 132 TF/s of useful work; the MFMA work alone is ~0.25 ms).  Here:
//   * 64 keys per iteration, TWO LDS stages: the global loads of iteration i + 1 (eight 16-byte loads per thread) are issued before
//     the arithmetic of iteration i, converted and written to the other stage after it — ONE barrier per 64 keys;
//   * 16-byte loads, pair-wise splits (split4m), V transposed in registers per (4 keys x 4 channels) item -> 8-byte LDS writes;
//   * 32x32x16 MFMAs: the wave's 32 queries are ONE tile (half the K / V^T fragment reads per MAC of the 16x16 form); S^T = K Q^T
//     leaves query (lane & 31) in the lane, and O^T = V^T P^T takes the lane's own exponentials as the column operand with the key
//     order of the accumulator layout (the V^T fragment is read in that order): no cross-lane traffic between the two products;
//   * online softmax with a LAZY reference maximum: the row's reference only moves when a block maximum exceeds it by more than 2^8
//     (scores are kept in log2 units: log2(e) folded into the Q scale), so the 64 accumulator multiplications of a rescale happen a
//     few times per row instead of once per block; p = exp2(s - ref) <= 256 stays well inside f16 / f32 range and the final
//     division by the row sum makes the result independent of the reference;
//   * each lane owns ONE query and 4 consecutive channels per accumulator quad: 16-byte f32 / 8-byte hi and lo stores.
__global__ __launch_bounds__(256, 2) void mha_mfma_f32_kernel(AttnArgs a) {
  constexpr int D = 64, KB = 64;
  constexpr int KROW = 144;                     // bytes per staged K row: 64 halves + 16 pad (16-byte fragment reads, conflict-free)
  constexpr int VROW = 136;                     // bytes per staged V^T row: 64 halves + 8 pad (8-byte fragment reads, conflict-free)
  constexpr int STAGE = 2 * KB * KROW + 2 * D * VROW;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
  const int g = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kg = lane >> 5;
  const float inv2048 = 1.0f / 2048.0f;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const float* Kb = (const float*)a.k + (long long)g * a.nk * a.ldk + a.koff + h * D;
  const float* Vb = (const float*)a.v + (long long)g * a.nk * a.ldv + a.voff + h * D;

  // ---- staging roles: K items (key = t16 + 16 j, channels d0..d0+3), j = 0..3; one V item (keys 4 t16 .. 4 t16 + 3, channels d0..d0+3)
  const int t16 = tid >> 4, d0 = (tid & 15) * 4;
  f32x4 kraw[4], vraw[4];
  auto load_block = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = k0 + t16 + 16 * j;
      kraw[j] = key < a.nk ? *reinterpret_cast<const f32x4*>(Kb + key * a.ldk + d0) : z4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + 4 * t16 + u;
      vraw[u] = key < a.nk ? *reinterpret_cast<const f32x4*>(Vb + key * a.ldv + d0) : z4;
    }
  };
  auto store_block = [&](unsigned char* st) {
    unsigned char* Kh = st;
    unsigned char* Kl = st + KB * KROW;
    unsigned char* Vh = st + 2 * KB * KROW;
    unsigned char* Vl = Vh + D * VROW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint2 kh, kl;
      split4m(kraw[j], kh, kl);
      *reinterpret_cast<uint2*>(Kh + (t16 + 16 * j) * KROW + d0 * 2) = kh;
      *reinterpret_cast<uint2*>(Kl + (t16 + 16 * j) * KROW + d0 * 2) = kl;
    }
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {                  // channel d0 + dd: its four keys 4 t16 .. 4 t16 + 3
      uint2 vh, vl;
      split4m(f32x4{vraw[0][dd], vraw[1][dd], vraw[2][dd], vraw[3][dd]}, vh, vl);
      *reinterpret_cast<uint2*>(Vh + (d0 + dd) * VROW + t16 * 8) = vh;
      *reinterpret_cast<uint2*>(Vl + (d0 + dd) * VROW + t16 * 8) = vl;
    }
  };
  load_block(0);

  // ---- Q fragments: query qi = q0 + col, channels 16 ks + 8 kg + 0..7, pre-multiplied by scale * log2(e)
  const int qi = blockIdx.x * 128 + wave * 32 + col;
  const float qscale = a.scale * 1.4426950408889634f;
  h16x8 qh[4], ql[4];
  {
    const float* qp = (const float*)a.q + ((long long)g * a.nq + min(qi, a.nq - 1)) * a.ldq + a.qoff + h * D + kg * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint2 h0, l0, h1, l1;
      split4m(*reinterpret_cast<const f32x4*>(qp + 16 * ks) * qscale, h0, l0);
      split4m(*reinterpret_cast<const f32x4*>(qp + 16 * ks + 4) * qscale, h1, l1);
      qh[ks] = __builtin_bit_cast(h16x8, u32x4{h0.x, h0.y, h1.x, h1.y});
      ql[ks] = __builtin_bit_cast(h16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
    }
  }
  f32x16 oM[2], oC[2];                                // O^T: channels dt * 32 + 8 (r >> 2) + 4 kg + (r & 3) of query col
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { oM[dt][e] = 0.f; oC[dt][e] = 0.f; }
  float mref = -INFINITY, lsum = 0.0f;                // reference maximum of the row (log2 units); this lane's part of the row sum

  store_block(lds);
  __syncthreads();
  const int nblk = (a.nk + KB - 1) / KB;
  for (int ib = 0; ib < nblk; ++ib) {
    const int k0 = ib * KB;
    unsigned char* st = lds + (ib & 1) * STAGE;
    if (ib + 1 < nblk) load_block(k0 + KB);           // in flight under this block's arithmetic
    const unsigned char* Kh = st;
    const unsigned char* Vh = st + 2 * KB * KROW;
    const bool tail = k0 + KB > a.nk;                 // uniform: only the last block can hold keys >= nk
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      // ---- S^T[key][query] for 32 keys: lane holds keys kt * 32 + (r & 3) + 8 (r >> 2) + 4 kg of query col
      f32x16 accM, accC;
#pragma unroll
      for (int e = 0; e < 16; ++e) { accM[e] = 0.f; accC[e] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const unsigned char* kr = Kh + (kt * 32 + col) * KROW + (16 * ks + 8 * kg) * 2;
        const h16x8 kh = *reinterpret_cast<const h16x8*>(kr);
        const h16x8 kl = *reinterpret_cast<const h16x8*>(kr + KB * KROW);
        accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], accM, 0, 0, 0);
        accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], accC, 0, 0, 0);
        accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], accC, 0, 0, 0);
      }
      f32x2 sc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[e] = f32x2{accC[2 * e], accC[2 * e + 1]} * inv2048 + f32x2{accM[2 * e], accM[2 * e + 1]};
      if (tail) {                                     // keys >= nk: -inf added (an addition, not a select: the result stays a canonical number
#pragma unroll                                        // for the maxima below)
        for (int e = 0; e < 8; ++e) {
          const int key = k0 + kt * 32 + ((2 * e) & 3) + 8 * (e >> 1) + 4 * kg;
          sc[e] += f32x2{key >= a.nk ? -INFINITY : 0.f, key + 1 >= a.nk ? -INFINITY : 0.f};
        }
      }
      float bm = fmaxf(sc[0][0], sc[0][1]);
#pragma unroll
      for (int e = 1; e < 8; ++e) bm = fmaxf(fmaxf(bm, sc[e][0]), sc[e][1]);     // v_max3_f32
      bm = fmaxf(bm, __shfl_xor(bm, 32));             // both halves of the wave hold the same queries
      const bool need = bm > mref + 8.0f;             // first block: mref = -inf; a fully masked sub-block (bm = -inf) never moves it
      if (__any(need)) {                              // wave-uniform branch, per-lane arithmetic
        const float mn = need ? bm : mref;
        const float corr = __builtin_amdgcn_exp2f(mref - mn);      // 1 where the reference stays; 0 on the first block
        mref = mn;
        lsum *= corr;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { oM[dt] *= corr; oC[dt] *= corr; }
      }
      f32x2 ps = {0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const f32x2 d = sc[e] - mref;
        sc[e] = f32x2{__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
        ps += sc[e];
      }
      lsum += ps[0] + ps[1];
      // ---- O^T += V^T P^T: k-step s covers the lane's values r = 8 s .. 8 s + 7 = keys 16 s + 4 kg + {0..3} and 16 s + 8 + 4 kg + {0..3}
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        unsigned ph[4], pl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2m(sc[4 * s + e][0], sc[4 * s + e][1], ph[e], pl[e]);
        const h16x8 phv = __builtin_bit_cast(h16x8, u32x4{ph[0], ph[1], ph[2], ph[3]});
        const h16x8 plv = __builtin_bit_cast(h16x8, u32x4{pl[0], pl[1], pl[2], pl[3]});
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned char* vr = Vh + (dt * 32 + col) * VROW + (kt * 32 + 16 * s + 4 * kg) * 2;
          const uint2 a0 = *reinterpret_cast<const uint2*>(vr), a1 = *reinterpret_cast<const uint2*>(vr + 16);
          const uint2 b0 = *reinterpret_cast<const uint2*>(vr + D * VROW), b1 = *reinterpret_cast<const uint2*>(vr + D * VROW + 16);
          const h16x8 vh = __builtin_bit_cast(h16x8, u32x4{a0.x, a0.y, a1.x, a1.y});
          const h16x8 vl = __builtin_bit_cast(h16x8, u32x4{b0.x, b0.y, b1.x, b1.y});
          oM[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, phv, oM[dt], 0, 0, 0);
          oC[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, phv, oC[dt], 0, 0, 0);
          oC[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, plv, oC[dt], 0, 0, 0);
        }
      }
    }
    if (ib + 1 < nblk) store_block(lds + ((ib + 1) & 1) * STAGE);   // the other stage: its last readers passed the barrier below one iteration ago
    __syncthreads();
  }
  // ---- normalise and store
  const float tot = lsum + __shfl_xor(lsum, 32);
  if (qi < a.nq) {
    const float inv = 1.0f / tot;
    const long long orow = (long long)g * a.nq + qi;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = (f32x4{oC[dt][4 * q], oC[dt][4 * q + 1], oC[dt][4 * q + 2], oC[dt][4 * q + 3]} * inv2048 +
                         f32x4{oM[dt][4 * q], oM[dt][4 * q + 1], oM[dt][4 * q + 2], oM[dt][4 * q + 3]}) * inv;
        const int c = a.ooff + h * D + dt * 32 + 8 * q + 4 * kg;
        if (a.osplit) {
          const float v[4] = {o[0], o[1], o[2], o[3]};
          uint2 hi, lo;
          omni_split4(v, hi, lo);
          unsigned char* p = (unsigned char*)a.o + orow * a.ldo * 4 + omni_split_off(c);
          *reinterpret_cast<uint2*>(p) = hi;
          *reinterpret_cast<uint2*>(p + 32) = lo;
        } else {
          *reinterpret_cast<f32x4*>((float*)a.o + orow * a.ldo + c) = o;
        }
      }
  }
}

// ------------------------------------------------------------------------------------ channel attention
struct ChanArgs {
  const void* qkv; void* o; float* ws;    // qkv [B*N, 3C]; ws [B][G][chunks][32][32]
  int B, N, C, G, chunks, chunk_tokens; float scale; int osplit;
};

template <typename T>
__global__ __launch_bounds__(256) void chan_scores_kernel(ChanArgs a) {
  // partial S[i][j] = sum_{n in chunk} q[n][i] * k[n][j]   (i, j in 0..31) for one (b, g, chunk)
  __shared__ float sq[64][33];
  __shared__ float sk[64][33];
  const int chunk = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const T* base = (const T*)a.qkv + (long long)b * a.N * 3 * a.C;
  const int i = threadIdx.x >> 3;            // 0..31
  const int j0 = (threadIdx.x & 7) * 4;      // 4 consecutive j
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int n0 = chunk * a.chunk_tokens;
  int n1 = n0 + a.chunk_tokens < a.N ? n0 + a.chunk_tokens : a.N;
  for (int t0 = n0; t0 < n1; t0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
      int tt = e >> 5, c = e & 31;
      int n = t0 + tt;
      float qv = 0.f, kv = 0.f;
      if (n < n1) {
        const T* row = base + (long long)n * 3 * a.C;
        qv = ldf(row + g * 32 + c);
        kv = ldf(row + a.C + g * 32 + c);
      }
      sq[tt][c] = qv; sk[tt][c] = kv;
    }
    __syncthreads();
#pragma unroll 8
    for (int tt = 0; tt < 64; ++tt) {
      float qv = sq[tt][i];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(qv, sk[tt][j0 + e], acc[e]);
    }
  }
  float* out = a.ws + ((((long long)b * a.G + g) * a.chunks + chunk) * 32 + i) * 32 + j0;
#pragma unroll
  for (int e = 0; e < 4; ++e) out[e] = acc[e];
}

template <typename T>
__global__ __launch_bounds__(256) void chan_apply_kernel(ChanArgs a) {
  // A = softmax_j(scale * sum_chunks S) ; out[n][g*32+i] = sum_j A[i][j] v[n][g*32+j]
  __shared__ float sa[32][33];
  const int g = blockIdx.y, b = blockIdx.z;
  {
    int i = threadIdx.x >> 3, j0 = (threadIdx.x & 7) * 4;
    const float* p = a.ws + (((long long)b * a.G + g) * a.chunks * 32 + i) * 32 + j0;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < a.chunks; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += p[(long long)c * 1024 + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) sa[i][j0 + e] = s[e] * a.scale;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    int i = threadIdx.x;
    float mx = -INFINITY;
    for (int j = 0; j < 32; ++j) mx = fmaxf(mx, sa[i][j]);
    float sum = 0.f;
    for (int j = 0; j < 32; ++j) { float e = expf(sa[i][j] - mx); sa[i][j] = e; sum += e; }
    float inv = 1.0f / sum;
    for (int j = 0; j < 32; ++j) sa[i][j] *= inv;
  }
  __syncthreads();
  int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= a.N) return;
  const T* vrow = (const T*)a.qkv + ((long long)b * a.N + n) * 3 * a.C + 2 * a.C + g * 32;
  float v[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) v[j] = ldf(vrow + j);
  T* orow = (T*)a.o + ((long long)b * a.N + n) * a.C + g * 32;
  unsigned char* srow = (unsigned char*)a.o + (((long long)b * a.N + n) * a.C) * 4;
#pragma unroll 2
  for (int i4 = 0; i4 < 32; i4 += 4) {
    float s4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) s = fmaf(sa[i4 + e][j], v[j], s);
      s4[e] = s;
    }
    if (a.osplit) {
      uint2 hi, lo;
      omni_split4(s4, hi, lo);
      unsigned char* p = srow + omni_split_off(g * 32 + i4);
      *reinterpret_cast<uint2*>(p) = hi;
      *reinterpret_cast<uint2*>(p + 32) = lo;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) stf(orow + i4 + e, s4[e]);
    }
  }
}

// Round-3 channel attention for f32 plans (three kernels; the pair above stays for f16 plans and as the A/B reference).
//   scores   S_partial[i][j] = sum_{n in chunk} q[n][i] k[n][j] on v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulation):
//            the MFMA's operand layout IS the memory layout — lane (i = lane & 31, t = lane >> 5) supplies q[n0 + t][i], so
//            both operands go from global memory straight into the matrix core, one dword per lane per step, with no LDS and no
//            VALU work.  The VALU kernel above read 20 B of LDS per 4 FMAs and was LDS-bound at 2.9 TB/s (profiles/r2_pmc_*).
//            Four waves per (chunk, group, image) take a quarter of the chunk each and are summed in wave order through LDS.
//   softmax  A = softmax_j(scale * sum_chunks S_partial) once per (image, group) — the apply kernel above recomputed it in every
//            256-token block (36 x 4 KB of partials per block in stage 0); A overwrites the group's chunk-0 slot of the workspace.
//   apply    out[n][i] = sum_j A[i][j] v[n][j], one token per thread: v as eight 16-byte loads (was 32 dword loads, each
//            touching 64 cache lines per wave), A rows 16-byte aligned in LDS.  Same fma chain per output as before.
__global__ __launch_bounds__(256) void chan_scores_mfma_kernel(ChanArgs a) {
  __shared__ float red[4][1024];
  const int chunk = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0c = chunk * a.chunk_tokens;
  const int n1 = n0c + a.chunk_tokens < a.N ? n0c + a.chunk_tokens : a.N;
  const int per = a.chunk_tokens / 4;                              // tokens per wave (chunk_tokens % 8 == 0)
  const int nw0 = n0c + wave * per, nw1 = min(nw0 + per, n1);
  const float* __restrict__ base = (const float*)a.qkv + (long long)b * a.N * 3 * a.C + g * 32 + (lane & 31);
  const long long ld = 3ll * a.C;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  const int tsel = lane >> 5;
  int n = nw0;
  for (; n + 32 <= nw1; n += 32) {                                 // 16 MFMA steps (32 tokens) per trip: 32 loads in flight
    const float* r = base + (long long)(n + tsel) * ld;
    float qv[16], kv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { qv[u] = r[2 * u * ld]; kv[u] = r[2 * u * ld + a.C]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[u], kv[u], acc, 0, 0, 0);
  }
  for (; n < nw1; n += 2) {                                        // ragged tail (small token counts)
    const int tok = n + tsel;
    const bool ok = tok < nw1;
    const float* r = base + (long long)(ok ? tok : nw0) * ld;
    const float qv = ok ? r[0] : 0.0f, kv = ok ? r[a.C] : 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qv, kv, acc, 0, 0, 0);
  }
  // D layout: col j = lane & 31, row i = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[e];
  __syncthreads();
  float* out = a.ws + (((long long)b * a.G + g) * a.chunks + chunk) * 1024;
  for (int e = threadIdx.x; e < 1024; e += 256) out[e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
}

__global__ __launch_bounds__(256) void chan_softmax_kernel(ChanArgs a) {
  __shared__ float sa[32][33];
  const int g = blockIdx.x, b = blockIdx.y;
  float* p0 = a.ws + ((long long)b * a.G + g) * a.chunks * 1024;
  const int i = threadIdx.x >> 3, j0 = (threadIdx.x & 7) * 4;
  {
    const float* p = p0 + i * 32 + j0;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < a.chunks; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += p[(long long)c * 1024 + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) sa[i][j0 + e] = s[e] * a.scale;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int r = threadIdx.x;
    float mx = -INFINITY;
    for (int j = 0; j < 32; ++j) mx = fmaxf(mx, sa[r][j]);
    float sum = 0.f;
    for (int j = 0; j < 32; ++j) { float e = expf(sa[r][j] - mx); sa[r][j] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < 32; ++j) sa[r][j] *= inv;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) p0[i * 32 + j0 + e] = sa[i][j0 + e];     // A over the chunk-0 partial of this (image, group)
}

// apply on the f16 matrix pipe (adopted in round 4: 2.55 vs 3.13 ms per launch at 4.7 M tokens x 128 channels, 3.5-3.8 TB/s instead
// of 2.9-3.1; profiles/r4_s2_per_op_{default,candidates}.txt — its predecessor, one token per lane with the 32x32 matrix broadcast from
// LDS, spent 1 024 v_fma_f32 + 256 ds_read_b128 per wave for 64 tokens and was bound by its vector ALU and the LDS pipe; an f32-MFMA
// variant (32x32x2, dword column loads) measured in round 3 was slower still).  Split-f16 x3 arithmetic of the attention kernels
// (hi * hi into one accumulator, hi * lo' + lo' * hi into a second, lo' = 2048 lo):
//   out^T[i][n] = sum_j A[i][j] v[n][j]:  the (image, group)'s 32x32 matrix is the ROW operand — split ONCE per wave into four
//   fragment registers sets; the token tile is the column operand — lane (n, kg) loads 2 x 32 contiguous bytes of token n's row
//   (16-byte loads, every byte used once) and splits its 16 values in registers; 6 MFMAs (32x32x16) per 32 tokens; the accumulator
//   quad q of a lane is 4 consecutive channels 8 q + 4 kg of ONE token: 16-byte f32 stores / 8-byte hi and lo stores.
// ~130 VALU instructions and 192 MFMA cycles per 32 tokens: a streaming kernel.
__global__ __launch_bounds__(256, 2) void chan_apply_mfma_split_kernel(ChanArgs a) {
  const int g = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, kg = lane >> 5;
  const float inv2048 = 1.0f / 2048.0f;
  // ---- all loads first: the wave's two 32-token tiles (rows clamped to the last token: tail tiles), then the matrix
  const int n00 = blockIdx.x * 256 + wave * 64;
  if (n00 >= a.N) return;                                    // wave-uniform
  const float* V = (const float*)a.qkv + (long long)b * a.N * 3 * a.C + 2 * a.C + g * 32 + kg * 8;
  f32x4 vraw[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int nl = min(n00 + t * 32 + col, a.N - 1);
    const float* vp = V + (long long)nl * 3 * a.C;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      vraw[t][2 * ks] = *reinterpret_cast<const f32x4*>(vp + 16 * ks);
      vraw[t][2 * ks + 1] = *reinterpret_cast<const f32x4*>(vp + 16 * ks + 4);
    }
  }
  // row operand: lane (i = col, kg) holds A[i][16 ks + 8 kg + 0..7]
  const float* A = a.ws + ((long long)b * a.G + g) * a.chunks * 1024 + col * 32 + kg * 8;
  h16x8 Ah[2], Al[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint2 h0, l0, h1, l1;
    split4m(*reinterpret_cast<const f32x4*>(A + 16 * ks), h0, l0);
    split4m(*reinterpret_cast<const f32x4*>(A + 16 * ks + 4), h1, l1);
    Ah[ks] = __builtin_bit_cast(h16x8, u32x4{h0.x, h0.y, h1.x, h1.y});
    Al[ks] = __builtin_bit_cast(h16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
  }
  float* const O = (float*)a.o + (long long)b * a.N * a.C + g * 32 + 4 * kg;
  unsigned char* const S = (unsigned char*)a.o + (long long)b * a.N * a.C * 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = n00 + t * 32 + col;
    if (n00 + t * 32 >= a.N) break;                          // wave-uniform
    f32x16 accM, accC;
#pragma unroll
    for (int e = 0; e < 16; ++e) { accM[e] = 0.f; accC[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint2 h0, l0, h1, l1;
      split4m(vraw[t][2 * ks], h0, l0);
      split4m(vraw[t][2 * ks + 1], h1, l1);
      const h16x8 vh = __builtin_bit_cast(h16x8, u32x4{h0.x, h0.y, h1.x, h1.y});
      const h16x8 vl = __builtin_bit_cast(h16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
      accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], vh, accM, 0, 0, 0);
      accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], vl, accC, 0, 0, 0);
      accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[ks], vh, accC, 0, 0, 0);
    }
    if (n < a.N) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {                          // channels 8 q + 4 kg + 0..3 of token n
        const f32x4 o = f32x4{accC[4 * q], accC[4 * q + 1], accC[4 * q + 2], accC[4 * q + 3]} * inv2048 +
                        f32x4{accM[4 * q], accM[4 * q + 1], accM[4 * q + 2], accM[4 * q + 3]};
        if (a.osplit) {
          const float v[4] = {o[0], o[1], o[2], o[3]};
          uint2 hi, lo;
          omni_split4(v, hi, lo);
          unsigned char* p = S + (long long)n * a.C * 4 + omni_split_off(g * 32 + 8 * q + 4 * kg);
          *reinterpret_cast<uint2*>(p) = hi;
          *reinterpret_cast<uint2*>(p + 32) = lo;
        } else {
          *reinterpret_cast<f32x4*>(O + (long long)n * a.C + 8 * q) = o;
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------ small glue kernels
struct PrepArgs { const void* x; const float* pos; const float* temporal; void* y; int B, N, C; };

template <typename T>
__global__ __launch_bounds__(256) void proj_prep_kernel(PrepArgs a) {
  // y[b][0][c] = mean_n(x[b][n][c] + pos[n][c]);  y[b][1+n][c] = x[b][n][c] + pos[n][c]
  int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (c >= a.C) return;
  const T* x = (const T*)a.x + (long long)b * a.N * a.C + c;
  T* y = (T*)a.y + (long long)b * (a.N + 1) * a.C + c;
  float s = 0.f;
  for (int n = 0; n < a.N; ++n) {
    float v = (ldf(x + (long long)n * a.C) + a.pos[(long long)n * a.C + c]) + a.temporal[c];
    s += v;
    stf(y + (long long)(n + 1) * a.C, v);
  }
  stf(y, s / (float)a.N);
}

struct AsmArgs { const void* img; const void* txt; void* y; int B, n_img, n_txt, C; };

template <typename T>
__global__ __launch_bounds__(256) void assemble_kernel(AsmArgs a) {
  long long total = (long long)a.B * (a.n_img + a.n_txt) * a.C;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int c = (int)(idx % a.C);
  long long r = idx / a.C;
  int t = (int)(r % (a.n_img + a.n_txt));
  long long b = r / (a.n_img + a.n_txt);
  const T* src = t < a.n_img ? (const T*)a.img + (b * a.n_img + t) * a.C + c : (const T*)a.txt + (long long)(t - a.n_img) * a.C + c;
  ((T*)a.y)[idx] = *src;
}

struct EmbArgs { const void* table; const void* pos; const int* ids; const int* step; void* y; int B, C, T, pos_offset; float scale; };

template <typename T>
__global__ __launch_bounds__(256) void embed_step_kernel(EmbArgs a) {
  // y[b] = table[ids[b][step]] * scale + pos[step + offset]
  int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (c >= a.C) return;
  int st = *a.step;
  int tok = a.ids[b * a.T + st];
  float v = ldf((const T*)a.table + (long long)tok * a.C + c);
  if (a.scale != 1.0f) v *= a.scale;
  v += ldf((const T*)a.pos + (long long)(st + a.pos_offset) * a.C + c);
  stf((T*)a.y + (long long)b * a.C + c, v);
}

// ------------------------------------------------------------------------------------ attn_decode
struct DecArgs {
  const void* q; const void* knew; const void* vnew; void* kc; void* vc; void* o; const int* step;
  int ldq, qoff, ldn, koff, voff, ldo;    // q [B, ldq]; new k/v rows in [B, ldn]
  int heads, nk_fixed, cap, C, ldc;       // cache [B, cap, ldc]; nk_fixed > 0 => cross attention over nk_fixed keys
  float scale;
};

template <typename T>
__global__ __launch_bounds__(64) void attn_decode_kernel(DecArgs a) {
  // one wave per (b, head), head_dim 64: lanes split keys for the scores, then split d for P.V
  OMNI_DYN_LDS(float, sp);                    // nk probabilities
  const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int C = a.ldc;
  int nk;
  T* Kc = (T*)a.kc + (long long)b * a.cap * C + h * 64;
  T* Vc = (T*)a.vc + (long long)b * a.cap * C + h * 64;
  if (a.nk_fixed > 0) {
    nk = a.nk_fixed;
  } else {
    int st = *a.step;
    nk = st + 1;
    // append this step's k, v (lane = d)
    Kc[(long long)st * C + lane] = ((const T*)a.knew)[(long long)b * a.ldn + a.koff + h * 64 + lane];
    Vc[(long long)st * C + lane] = ((const T*)a.vnew)[(long long)b * a.ldn + a.voff + h * 64 + lane];
    __syncthreads();
  }
  float qd = ldf((const T*)a.q + (long long)b * a.ldq + a.qoff + h * 64 + lane);
  __shared__ float sq[64];
  sq[lane] = qd;
  __syncthreads();
  float mx = -INFINITY;
  for (int k = lane; k < nk; k += 64) {
    const T* kr = Kc + (long long)k * C;
    float s = 0.f;
#pragma unroll 16
    for (int d = 0; d < 64; ++d) s += sq[d] * ldf(kr + d);
    s *= a.scale;
    sp[k] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < nk; k += 64) { float e = expf(sp[k] - mx); sp[k] = e; sum += e; }
  sum = wave_sum(sum);
  __syncthreads();
  float acc = 0.f;
  for (int k = 0; k < nk; ++k) acc += sp[k] * ldf(Vc + (long long)k * C + lane);
  stf((T*)a.o + (long long)b * a.ldo + h * 64 + lane, acc / sum);
}

// Cross-attention of a decode step (nk_fixed keys, f32 plans), round 3.  The kernel above gives every lane its own K row (64 dword
// loads per lane, 64 cache lines per load instruction) and runs one wave per (row, head): 1.5 waves per SIMD at 128 crops, 2.5 TB/s
// on the 460 MB cross-KV of a step.  Here a block of four waves serves one (row, head); each wave takes a quarter of the keys and
// reads FOUR keys per load instruction as four contiguous 256-byte rows (lane = (key j = lane >> 4, 16-byte slot lane & 15)), the
// 16 lanes of a key reduce their partial dot products with four xor-shuffles, eight such loads are in flight per lane.  P.V uses
// the same mapping (a lane accumulates its slot over its keys, the four key lanes are folded at the end) and the four waves are
// summed in wave order through LDS.  Same softmax (expf, max-subtracted), different summation order than the kernel above.
__global__ __launch_bounds__(256) void attn_decode_cross_kernel(DecArgs a) {
  OMNI_DYN_LDS(float, sp);                    // [nk] scores -> probabilities, then 8 reduction slots, then [4][64] partial outputs
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane >> 4, d4 = (lane & 15) * 4;
  const int nk = a.nk_fixed, C = a.ldc;
  float* red = sp + ((nk + 3) & ~3);
  float* part = red + 8;
  const float* __restrict__ Kc = (const float*)a.kc + (long long)b * a.cap * C + h * 64 + d4;
  const float* __restrict__ Vc = (const float*)a.vc + (long long)b * a.cap * C + h * 64 + d4;
  const f32x4 q = *reinterpret_cast<const f32x4*>((const float*)a.q + (long long)b * a.ldq + a.qoff + h * 64 + d4);
  const int per = ((nk + 15) >> 4) << 2;      // keys per wave, a multiple of 4
  const int k0 = wave * per, k1 = min(k0 + per, nk);
  float mx = -INFINITY;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int kb = k0; kb < k1; kb += 32) {      // eight 4-key loads in flight; all 64 lanes stay in the loop for the shuffles
    f32x4 kv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = kb + 4 * u + j;
      kv[u] = k < k1 ? *reinterpret_cast<const f32x4*>(Kc + (long long)k * C) : z4;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = kb + 4 * u + j;
      float s = (q[0] * kv[u][0] + q[1] * kv[u][1]) + (q[2] * kv[u][2] + q[3] * kv[u][3]);
      s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
      s *= a.scale;
      if (k < k1) {
        if ((lane & 15) == 0) sp[k] = s;
        mx = fmaxf(mx, s);
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int k = tid; k < nk; k += 256) { const float e = expf(sp[k] - mx); sp[k] = e; sum += e; }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = ((red[4] + red[5]) + red[6]) + red[7];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kb = k0; kb < k1; kb += 32) {
    f32x4 vv[8];
    float pk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = kb + 4 * u + j;
      const bool ok = k < k1;
      vv[u] = ok ? *reinterpret_cast<const f32x4*>(Vc + (long long)k * C) : z4;
      pk[u] = ok ? sp[k] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] += pk[u] * vv[u][0]; acc[1] += pk[u] * vv[u][1]; acc[2] += pk[u] * vv[u][2]; acc[3] += pk[u] * vv[u][3];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { acc[e] += __shfl_xor(acc[e], 16); acc[e] += __shfl_xor(acc[e], 32); }
  if (lane < 16) *reinterpret_cast<f32x4*>(part + wave * 64 + d4) = acc;
  __syncthreads();
  if (tid < 64) {
    const float o = ((part[tid] + part[64 + tid]) + part[128 + tid]) + part[192 + tid];
    ((float*)a.o)[(long long)b * a.ldo + h * 64 + tid] = o / sum;
  }
}

// ------------------------------------------------------------------------------------ greedy_step
struct GreedyArgs {
  const void* logits; const float* bias; int* ids; int* finished; const int* step;
  int B, V, ldl, T, max_new, ngram, bos, eos, pad, forced_bos, forced_eos;
};

template <typename T>
__global__ __launch_bounds__(256) void greedy_step_kernel(GreedyArgs a) {
  __shared__ float sval[256];
  __shared__ int sidx[256];
  __shared__ int banned[32];
  __shared__ int nban;
  const int b = blockIdx.x;
  const int st = *a.step;              // tokens so far = st + 1 (ids[b][0..st]); we write ids[b][st+1]
  int* ids = a.ids + b * a.T;
  const int cur_len = st + 1;
  if (threadIdx.x == 0) {
    int nb = 0;
    // NoRepeatNGram: ban tokens completing an n-gram already generated (incl. decoder start token)
    if (a.ngram > 0 && cur_len + 1 >= a.ngram) {
      int n = a.ngram;
      for (int i = 0; i + n - 1 < cur_len; ++i) {
        bool match = true;
        for (int j = 0; j < n - 1; ++j)
          if (ids[i + j] != ids[cur_len - (n - 1) + j]) { match = false; break; }
        if (match && nb < 32) banned[nb++] = ids[i + n - 1];
      }
    }
    nban = nb;
  }
  __syncthreads();
  int forced = -1;
  if (a.forced_bos >= 0 && cur_len == 1) forced = a.forced_bos;
  if (a.forced_eos >= 0 && cur_len == a.max_new) forced = a.forced_eos;   // max_length - 1 == max_new (start token + max_new)
  int tok;
  if (forced >= 0) {
    tok = forced;
  } else {
    const T* lg = (const T*)a.logits + (long long)b * a.ldl;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < a.V; v += 256) {
      float x = ldf(lg + v) + (a.bias ? a.bias[v] : 0.0f);
      for (int q = 0; q < nban; ++q) if (banned[q] == v) x = -INFINITY;
      if (x > best) { best = x; bi = v; }      // first max within this thread's ascending stride
    }
    sval[threadIdx.x] = best; sidx[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        float ov = sval[threadIdx.x + s]; int oi = sidx[threadIdx.x + s];
        if (ov > sval[threadIdx.x] || (ov == sval[threadIdx.x] && oi < sidx[threadIdx.x])) {
          sval[threadIdx.x] = ov; sidx[threadIdx.x] = oi;
        }
      }
      __syncthreads();
    }
    tok = sidx[0];
    // no logit compared greater than -inf: every entry is NaN or -inf (a row nobody asked for, computed from padding, or a diverged
    // input).  torch.argmax returns position 0 then; never hand an out-of-range id to the next step's embedding gather
    if (tok < 0 || tok >= a.V) tok = 0;
  }
  if (threadIdx.x == 0) {
    int fin = a.finished[b];
    if (fin) tok = a.pad;                       // finished rows keep emitting pad (hf utils.py:2925-2929)
    ids[st + 1] = tok;
    if (!fin && tok == a.eos) a.finished[b] = 1;
  }
}

__global__ void step_inc_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }

// ------------------------------------------------------------------------------------ crop_resize
struct CropArgs {
  const unsigned char* img; const int* boxes; unsigned char* c64; unsigned char* tmp; const int* b; const int* k;
  void* y; const float* lut; int n, H, W, R, ksize, ldo; float mean[3], stdv[3];
};

__device__ __forceinline__ int cv_round_short(float v) {   // saturate_cast<short>(float): round half to even
  float r = rintf(v);
  r = fminf(fmaxf(r, -32768.f), 32767.f);
  return (int)r;
}

__global__ __launch_bounds__(256) void crop_bilinear64_kernel(CropArgs a) {
  // OpenCV resize INTER_LINEAR, 8UC3, fixed point (INTER_RESIZE_COEF_BITS = 11): SURVEY App. A.2
  int idx = blockIdx.x * 256 + threadIdx.x;
  int crop = blockIdx.y;
  if (idx >= 64 * 64) return;
  int dy = idx >> 6, dx = idx & 63;
  int x0 = a.boxes[crop * 4 + 0], y0 = a.boxes[crop * 4 + 1], x1 = a.boxes[crop * 4 + 2], y1 = a.boxes[crop * 4 + 3];
  // numpy slicing image_source[ymin:ymax, xmin:xmax] (ref:util/utils.py:99) clips the rectangle to the image; a rectangle that is
  // empty after clipping (the reference's cv2.resize raises on it) gives a black crop — the kernel never reads outside the frame
  x0 = min(max(x0, 0), a.W); x1 = min(max(x1, 0), a.W); y0 = min(max(y0, 0), a.H); y1 = min(max(y1, 0), a.H);
  int sw = x1 - x0, sh = y1 - y0;
  unsigned char* dst = a.c64 + ((long long)crop * 4096 + idx) * 3;
  if (sw <= 0 || sh <= 0) { dst[0] = dst[1] = dst[2] = 0; return; }
  double scale_x = (double)sw / 64.0, scale_y = (double)sh / 64.0;
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = (int)floorf(fx); fx -= sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
  // vertical: beta from the un-clamped fraction, source rows clipped to [0, sh-1] (cv::resize invoker)
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  int sy = (int)floorf(fy); fy -= sy;
  int a0 = cv_round_short((1.f - fx) * 2048.f), a1 = cv_round_short(fx * 2048.f);
  int b0 = cv_round_short((1.f - fy) * 2048.f), b1 = cv_round_short(fy * 2048.f);
  int sx1 = sx + 1 < sw ? sx + 1 : sx;
  int sy0c = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
  int sy1c = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
  const unsigned char* r0 = a.img + ((long long)(y0 + sy0c) * a.W + x0) * 3;
  const unsigned char* r1 = a.img + ((long long)(y0 + sy1c) * a.W + x0) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int S0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
    int S1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
    int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    dst[c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

__device__ __forceinline__ int clip8b(int v) { v >>= 22; return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void bicubic_h_kernel(CropArgs a) {
  // 64 -> R horizontal pass (Pillow 8bpc), per crop: tmp [n][64][R][3]
  int idx = blockIdx.x * 256 + threadIdx.x, crop = blockIdx.y;
  if (idx >= 64 * a.R) return;
  int y = idx / a.R, xx = idx - y * a.R;
  int xmin = a.b[xx * 2], cnt = a.b[xx * 2 + 1];
  const int* k = a.k + (long long)xx * a.ksize;
  const unsigned char* src = a.c64 + ((long long)crop * 4096 + y * 64 + xmin) * 3;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int x = 0; x < cnt; ++x) { int kk = k[x]; s0 += src[x * 3] * kk; s1 += src[x * 3 + 1] * kk; s2 += src[x * 3 + 2] * kk; }
  unsigned char* d = a.tmp + ((long long)crop * 64 * a.R + idx) * 3;
  d[0] = (unsigned char)clip8b(s0); d[1] = (unsigned char)clip8b(s1); d[2] = (unsigned char)clip8b(s2);
}

template <typename T>
__global__ __launch_bounds__(256) void bicubic_v_norm_kernel(CropArgs a) {
  // vertical pass (same square tables) + rescale 1/255 + (x - mean) / std  ->  y [n][R][R][ldo]
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  int crop = blockIdx.y;
  if (idx >= (long long)a.R * a.R) return;
  int yy = (int)(idx / a.R), xx = (int)(idx - (long long)yy * a.R);
  int p[3];
  if (a.R == 64) {
    const unsigned char* s = a.c64 + ((long long)crop * 4096 + idx) * 3;
    p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
  } else {
    int ymin = a.b[yy * 2], cnt = a.b[yy * 2 + 1];
    const int* k = a.k + (long long)yy * a.ksize;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int y = 0; y < cnt; ++y) {
      const unsigned char* s = a.tmp + (((long long)crop * 64 + ymin + y) * a.R + xx) * 3;
      int kk = k[y]; s0 += s[0] * kk; s1 += s[1] * kk; s2 += s[2] * kk;
    }
    p[0] = clip8b(s0); p[1] = clip8b(s1); p[2] = clip8b(s2);
  }
  T* out = (T*)a.y + ((long long)crop * a.R * a.R + idx) * a.ldo;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = a.lut[p[c]];      // f32(f64(p) * (1/255)): hf image_transforms.rescale, host-built table
    stf(out + c, (v - a.mean[c]) / a.stdv[c]);
  }
  for (int c = 3; c < a.ldo; ++c) stf(out + c, 0.0f);
}

template <typename F32, typename F16>
int by_dtype(int dtype, const char* name, F32 f32, F16 f16) {
  if (dtype == OMNI_F32) f32();
  else if (dtype == OMNI_F16) f16();
  else { omni_set_error("%s: bad dtype %d", name, dtype); return OMNI_E_ARG; }
  return OMNI_OK;
}

}  // namespace

int omni_launch_dwconv3(const omni_op_t* op, hipStream_t s) {
  DwArgs a{};
  a.x = op->p[0]; a.w = op->p[1]; a.bias = (const float*)op->p[2]; a.y = op->p[4];
  a.B = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.C = op->i[3];
  OMNI_REQUIRE(a.x && a.w && a.bias && a.y && a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0, "dwconv3: bad arguments");
  OMNI_REQUIRE(a.C % (op->dtype == OMNI_F32 ? 4 : 8) == 0, "dwconv3: C must be a multiple of the 16-byte vector width");
  a.total = (long long)a.B * a.H * a.W * a.C;
  const int V = op->dtype == OMNI_F32 ? 4 : 8;
  const DwStripGrid g = dwconv3_strip_grid(a, V);
  int rc;
  if (g.ok) {
    rc = by_dtype(op->dtype, "dwconv3",
        [&] { hipLaunchKernelGGL(dwconv3_strip_kernel<float>, dim3(g.gx, g.gy, g.gz), dim3(256), 0, s, a, g.cv_log2); },
        [&] { hipLaunchKernelGGL(dwconv3_strip_kernel<half_t>, dim3(g.gx, g.gy, g.gz), dim3(256), 0, s, a, g.cv_log2); });
  } else {
    long long blocks = (a.total / V + 255) / 256; if (blocks > 65536) blocks = 65536;
    rc = by_dtype(op->dtype, "dwconv3",
        [&] { hipLaunchKernelGGL(dwconv3_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a); },
        [&] { hipLaunchKernelGGL(dwconv3_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, s, a); });
  }
  if (rc) return rc;
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

int omni_launch_dwconv3_ln(const omni_op_t* op, hipStream_t s) {
  DwLnArgs a{};
  a.x = op->p[0]; a.w = op->p[1]; a.bias = (const float*)op->p[2]; a.h = op->p[3]; a.y1 = op->p[4];
  a.g = (const float*)op->p[5]; a.b = (const float*)op->p[6];
  a.B = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.C = op->i[3]; a.eps = op->f[0]; a.osplit = op->i[6];
  const int V = op->dtype == OMNI_F32 ? 4 : 8;
  OMNI_REQUIRE(a.x && a.w && a.bias && a.h && a.y1 && a.g && a.b, "dwconv3_ln: null pointer");
  OMNI_REQUIRE(!a.osplit || (op->dtype == OMNI_F32 && a.C % 16 == 0), "dwconv3_ln: split output needs an f32 plan and C %% 16 == 0");
  OMNI_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0 && a.C <= 1024 && a.C % V == 0, "dwconv3_ln: bad shape (C <= 1024, C %% %d == 0)", V);
  a.pixels = (long long)a.B * a.H * a.W;
  if (op->dtype == OMNI_F32 && (a.C == 128 || a.C == 256 || a.C == 512) && (long long)a.B * a.H * a.W < (1ll << 31)) {
    if (a.C == 128) launch_dwln_strip<1, 32>(a, s);
    else if (a.C == 256) launch_dwln_strip<1, 64>(a, s);
    else launch_dwln_strip<2, 64>(a, s);
    OMNI_HIP_CHECK(hipGetLastError());
    return OMNI_OK;
  }
  unsigned blocks = (unsigned)((a.pixels + 3) / 4);
  int rc = by_dtype(op->dtype, "dwconv3_ln",
      [&] { hipLaunchKernelGGL(dwconv3_ln_kernel<float>, dim3(blocks), dim3(256), 0, s, a); },
      [&] { hipLaunchKernelGGL(dwconv3_ln_kernel<half_t>, dim3(blocks), dim3(256), 0, s, a); });
  if (rc) return rc;
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

int omni_launch_layernorm(const omni_op_t* op, hipStream_t s) {
  LnArgs a{};
  a.x = op->p[0]; a.add = op->p[1]; a.g = (const float*)op->p[2]; a.b = (const float*)op->p[3]; a.y = op->p[4];
  a.rows = ((long long)op->i[0]) * (op->i[1] > 0 ? op->i[1] : 1); a.C = op->i[3]; a.period = op->i[5] > 0 ? op->i[5] : 1;
  a.eps = op->f[0];
  a.omode = op->i[6]; a.y2 = op->p[5];
  OMNI_REQUIRE(a.x && a.g && a.b && a.y && a.rows > 0 && a.C > 0 && a.C <= 1024, "layernorm: bad arguments (C <= 1024)");
  OMNI_REQUIRE(a.omode >= 0 && a.omode <= 2 && (a.omode != 2 || a.y2), "layernorm: bad output mode %d", a.omode);
  OMNI_REQUIRE(a.omode == 0 || (op->dtype == OMNI_F32 && a.C % 16 == 0), "layernorm: split output needs an f32 plan and C %% 16 == 0");
  if (op->dtype == OMNI_F32 && launch_layernorm_v4(a, s)) {
    OMNI_HIP_CHECK(hipGetLastError());
    return OMNI_OK;
  }
  unsigned blocks = (unsigned)((a.rows + 3) / 4);
  int rc = by_dtype(op->dtype, "layernorm",
      [&] { launch_layernorm_typed<float>(a, blocks, s); },
      [&] { launch_layernorm_typed<half_t>(a, blocks, s); });
  if (rc) return rc;
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

static int launch_attn_rows(const omni_op_t* op, hipStream_t s) {
  AttnArgs a{};
  a.q = op->p[0]; a.k = op->p[1]; a.v = op->p[2]; a.o = op->p[4];
  a.kbias = (const float*)op->p[5]; a.vbias = (const float*)op->p[6];
  a.ldq = op->i[0]; a.ldk = op->i[1]; a.ldv = op->i[2]; a.ldo = op->i[3];
  a.qoff = op->i[4]; a.koff = op->i[5]; a.voff = op->i[6]; a.ooff = op->i[7];
  a.heads = op->i[8]; a.nq = op->i[9]; a.nk = op->i[10]; a.groups = op->i[11];
  a.mode = op->i[12]; a.H = op->i[13]; a.W = op->i[14];
  int D = op->i[15];
  a.scale = op->f[0];
  a.osplit = op->i[16];
  OMNI_REQUIRE(a.q && a.k && a.v && a.o && a.heads > 0 && a.nq > 0 && a.nk > 0 && a.groups > 0, "attn_rows: bad arguments");
  OMNI_REQUIRE(!a.osplit || (op->dtype == OMNI_F32 && a.ldo % 16 == 0 && a.ooff % 16 == 0), "attn_rows: split output needs an f32 plan, 16-channel aligned");
  OMNI_REQUIRE(D == 32 || D == 64, "attn_rows: head_dim %d unsupported", D);
  if (a.mode == 1) {
    OMNI_REQUIRE(a.nq == 144 && a.nk == 144 && a.H > 0 && a.W > 0, "attn_rows: window mode needs 12x12 windows");
    a.wy = (a.H + 11) / 12; a.wx = (a.W + 11) / 12;
    OMNI_REQUIRE(a.groups % (a.wy * a.wx) == 0, "attn_rows: groups must be B * windows");
  } else { a.wy = a.wx = 0; }
  int rc;
  const bool aligned4 = a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0 &&
                        a.qoff % 4 == 0 && a.koff % 4 == 0 && a.voff % 4 == 0 && a.ooff % 4 == 0;
  OMNI_REQUIRE(!a.osplit || (a.mode == 1 && D == 32) || (a.mode == 0 && D == 64), "attn_rows: split output exists on the MFMA kernels only");
  if (a.mode == 1 && D == 32) {       // 12x12 window attention on the matrix cores (split-f16)
    dim3 grid(1, a.heads, a.groups);
    OMNI_REQUIRE(op->dtype != OMNI_F32 || aligned4, "attn_rows: the f32 window kernel reads 16-byte rows (pitches and channel offsets %% 4 == 0)");
    rc = by_dtype(op->dtype, "attn_rows",
      [&] { hipLaunchKernelGGL(window_attn_mfma_f32_kernel, grid, dim3(192), 0, s, a); },
      [&] { hipLaunchKernelGGL((window_attn_mfma_kernel<half_t>), grid, dim3(192), 0, s, a); });
  } else if (a.mode == 0 && D == 64) {   // BART encoder MHA on the matrix cores (flash-style, split-f16)
    dim3 grid((a.nq + 127) / 128, a.heads, a.groups);
    OMNI_REQUIRE(op->dtype != OMNI_F32 || aligned4, "attn_rows: the f32 MHA kernel reads 16-byte rows (pitches and channel offsets %% 4 == 0)");
    rc = by_dtype(op->dtype, "attn_rows",
      [&] { hipLaunchKernelGGL(mha_mfma_f32_kernel, grid, dim3(256), 0, s, a); },
      [&] { hipLaunchKernelGGL((mha_mfma_kernel<half_t>), grid, dim3(256), 0, s, a); });
  } else {
    dim3 grid((a.nq + 127) / 128, a.heads, a.groups);
    if (D == 32) rc = by_dtype(op->dtype, "attn_rows",
      [&] { hipLaunchKernelGGL((attn_rows_kernel<float, 32, 128>), grid, dim3(128), 0, s, a); },
      [&] { hipLaunchKernelGGL((attn_rows_kernel<half_t, 32, 128>), grid, dim3(128), 0, s, a); });
    else rc = by_dtype(op->dtype, "attn_rows",
      [&] { hipLaunchKernelGGL((attn_rows_kernel<float, 64, 128>), grid, dim3(128), 0, s, a); },
      [&] { hipLaunchKernelGGL((attn_rows_kernel<half_t, 64, 128>), grid, dim3(128), 0, s, a); });
  }
  if (rc) return rc;
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

static int launch_chan_attn(const omni_op_t* op, hipStream_t s) {
  ChanArgs a{};
  a.qkv = op->p[0]; a.o = op->p[4]; a.ws = (float*)op->p[5];
  a.B = op->i[0]; a.N = op->i[1]; a.C = op->i[3]; a.G = op->i[4]; a.chunk_tokens = op->i[5]; a.osplit = op->i[6];
  OMNI_REQUIRE(a.qkv && a.o && a.ws && a.B > 0 && a.N > 0 && a.C == a.G * 32 && a.chunk_tokens > 0, "chan_attn: bad arguments");
  OMNI_REQUIRE(!a.osplit || op->dtype == OMNI_F32, "chan_attn: split output needs an f32 plan");
  a.chunks = (a.N + a.chunk_tokens - 1) / a.chunk_tokens;
  a.scale = 1.0f / sqrtf((float)a.N);
  if (op->f[0] != 0.0f) a.scale = op->f[0];
  dim3 g1(a.chunks, a.G, a.B), g2((a.N + 255) / 256, a.G, a.B);
  if (op->dtype == OMNI_F32 && a.chunk_tokens % 8 == 0 && a.C % 4 == 0) {
    hipLaunchKernelGGL(chan_scores_mfma_kernel, g1, dim3(256), 0, s, a);
    hipLaunchKernelGGL(chan_softmax_kernel, dim3(a.G, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(chan_apply_mfma_split_kernel, g2, dim3(256), 0, s, a);
    OMNI_HIP_CHECK(hipGetLastError());
    return OMNI_OK;
  }
  int rc = by_dtype(op->dtype, "chan_attn",
      [&] { hipLaunchKernelGGL(chan_scores_kernel<float>, g1, dim3(256), 0, s, a);
            hipLaunchKernelGGL(chan_apply_kernel<float>, g2, dim3(256), 0, s, a); },
      [&] { hipLaunchKernelGGL(chan_scores_kernel<half_t>, g1, dim3(256), 0, s, a);
            hipLaunchKernelGGL(chan_apply_kernel<half_t>, g2, dim3(256), 0, s, a); });
  if (rc) return rc;
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

static int launch_attn_decode(const omni_op_t* op, hipStream_t s) {
  DecArgs a{};
  a.q = op->p[0]; a.knew = op->p[1]; a.vnew = op->p[2]; a.kc = op->p[3]; a.o = op->p[4]; a.vc = op->p[5];
  a.step = (const int*)op->p[6];
  a.ldq = op->i[0]; a.qoff = op->i[1]; a.ldn = op->i[2]; a.koff = op->i[3]; a.voff = op->i[4]; a.ldo = op->i[5];
  a.heads = op->i[6]; a.nk_fixed = op->i[7]; a.cap = op->i[8]; a.C = op->i[9]; a.ldc = op->i[11] > 0 ? op->i[11] : a.C;
  int B = op->i[10];
  a.scale = op->f[0];
  OMNI_REQUIRE(a.q && a.kc && a.vc && a.o && B > 0 && a.heads > 0 && a.C == a.heads * 64, "attn_decode: bad arguments (head_dim 64)");
  OMNI_REQUIRE(a.nk_fixed > 0 || (a.knew && a.vnew && a.step), "attn_decode: self-attention needs new k/v and the step counter");
  int nk_max = a.nk_fixed > 0 ? a.nk_fixed : a.cap;
  if (a.nk_fixed > 0 && op->dtype == OMNI_F32 && a.ldc % 4 == 0 && a.ldq % 4 == 0 && a.qoff % 4 == 0) {
    // cross-attention over the fixed encoder keys: four waves per (row, head), four keys per load instruction
    const size_t lds = (size_t)(((a.nk_fixed + 3) & ~3) + 8 + 256) * 4;
    hipLaunchKernelGGL(attn_decode_cross_kernel, dim3(a.heads, B), dim3(256), lds, s, a);
    OMNI_HIP_CHECK(hipGetLastError());
    return OMNI_OK;
  }
  dim3 grid(a.heads, B);
  size_t sh = (size_t)nk_max * 4;
  int rc = by_dtype(op->dtype, "attn_decode",
      [&] { hipLaunchKernelGGL(attn_decode_kernel<float>, grid, dim3(64), sh, s, a); },
      [&] { hipLaunchKernelGGL(attn_decode_kernel<half_t>, grid, dim3(64), sh, s, a); });
  if (rc) return rc;
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

static int launch_greedy(const omni_op_t* op, hipStream_t s) {
  GreedyArgs a{};
  a.logits = op->p[0]; a.bias = (const float*)op->p[1]; a.ids = (int*)op->p[2]; a.finished = (int*)op->p[3];
  a.step = (const int*)op->p[6];
  a.B = op->i[0]; a.V = op->i[1]; a.ldl = op->i[2]; a.T = op->i[3]; a.max_new = op->i[4]; a.ngram = op->i[5];
  a.bos = op->i[6]; a.eos = op->i[7]; a.pad = op->i[8]; a.forced_bos = op->i[9]; a.forced_eos = op->i[10];
  OMNI_REQUIRE(a.logits && a.ids && a.finished && a.step && a.B > 0 && a.V > 0 && a.T >= a.max_new + 1, "greedy_step: bad arguments");
  int rc = by_dtype(op->dtype, "greedy_step",
      [&] { hipLaunchKernelGGL(greedy_step_kernel<float>, dim3(a.B), dim3(256), 0, s, a); },
      [&] { hipLaunchKernelGGL(greedy_step_kernel<half_t>, dim3(a.B), dim3(256), 0, s, a); });
  if (rc) return rc;
  if (op->i[11]) hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, s, (int*)op->p[6]);
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

static int launch_crop_resize(const omni_op_t* op, hipStream_t s) {
  CropArgs a{};
  a.img = (const unsigned char*)op->p[0]; a.boxes = (const int*)op->p[1]; a.c64 = (unsigned char*)op->p[2];
  a.tmp = (unsigned char*)op->p[3]; a.y = op->p[4]; a.b = (const int*)op->p[5]; a.k = (const int*)op->p[6];
  a.lut = (const float*)op->p[7];
  a.n = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.R = op->i[3]; a.ksize = op->i[4]; a.ldo = op->i[13];
  for (int c = 0; c < 3; ++c) { a.mean[c] = op->f[c]; a.stdv[c] = op->f[3 + c]; }
  OMNI_REQUIRE(a.img && a.boxes && a.c64 && a.y && a.lut && a.n > 0 && a.H > 0 && a.W > 0 && a.R >= 64 && a.ldo >= 3, "crop_resize: bad arguments");
  OMNI_REQUIRE(a.R == 64 || (a.tmp && a.b && a.k && a.ksize > 0), "crop_resize: bicubic tables missing");
  hipLaunchKernelGGL(crop_bilinear64_kernel, dim3(16, a.n), dim3(256), 0, s, a);
  if (a.R != 64) hipLaunchKernelGGL(bicubic_h_kernel, dim3((64 * a.R + 255) / 256, a.n), dim3(256), 0, s, a);
  dim3 g((unsigned)(((long long)a.R * a.R + 255) / 256), a.n);
  int rc = by_dtype(op->dtype, "crop_resize",
      [&] { hipLaunchKernelGGL(bicubic_v_norm_kernel<float>, g, dim3(256), 0, s, a); },
      [&] { hipLaunchKernelGGL(bicubic_v_norm_kernel<half_t>, g, dim3(256), 0, s, a); });
  if (rc) return rc;
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

static int launch_glue(const omni_op_t* op, hipStream_t s) {
  int rc = OMNI_OK;
  if (op->kind == OMNI_OP_PROJ_PREP) {
    PrepArgs a{}; a.x = op->p[0]; a.pos = (const float*)op->p[1]; a.temporal = (const float*)op->p[2]; a.y = op->p[4]; a.B = op->i[0]; a.N = op->i[1]; a.C = op->i[3];
    OMNI_REQUIRE(a.x && a.pos && a.temporal && a.y && a.B > 0 && a.N > 0 && a.C > 0, "proj_prep: bad arguments");
    dim3 g((a.C + 255) / 256, a.B);
    rc = by_dtype(op->dtype, "proj_prep",
        [&] { hipLaunchKernelGGL(proj_prep_kernel<float>, g, dim3(256), 0, s, a); },
        [&] { hipLaunchKernelGGL(proj_prep_kernel<half_t>, g, dim3(256), 0, s, a); });
  } else if (op->kind == OMNI_OP_ASSEMBLE) {
    AsmArgs a{}; a.img = op->p[0]; a.txt = op->p[1]; a.y = op->p[4]; a.B = op->i[0]; a.n_img = op->i[1]; a.n_txt = op->i[2]; a.C = op->i[3];
    OMNI_REQUIRE(a.img && a.txt && a.y && a.B > 0 && a.n_img > 0 && a.n_txt >= 0 && a.C > 0, "assemble: bad arguments");
    long long total = (long long)a.B * (a.n_img + a.n_txt) * a.C;
    dim3 g((unsigned)((total + 255) / 256));
    rc = by_dtype(op->dtype, "assemble",
        [&] { hipLaunchKernelGGL(assemble_kernel<float>, g, dim3(256), 0, s, a); },
        [&] { hipLaunchKernelGGL(assemble_kernel<half_t>, g, dim3(256), 0, s, a); });
  } else {
    EmbArgs a{}; a.table = op->p[0]; a.pos = op->p[1]; a.ids = (const int*)op->p[2]; a.y = op->p[4]; a.step = (const int*)op->p[6];
    a.B = op->i[0]; a.C = op->i[3]; a.T = op->i[4]; a.pos_offset = op->i[5]; a.scale = op->f[0] == 0.0f ? 1.0f : op->f[0];
    OMNI_REQUIRE(a.table && a.pos && a.ids && a.y && a.step && a.B > 0 && a.C > 0, "embed_step: bad arguments");
    dim3 g((a.C + 255) / 256, a.B);
    rc = by_dtype(op->dtype, "embed_step",
        [&] { hipLaunchKernelGGL(embed_step_kernel<float>, g, dim3(256), 0, s, a); },
        [&] { hipLaunchKernelGGL(embed_step_kernel<half_t>, g, dim3(256), 0, s, a); });
  }
  if (rc) return rc;
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

int omni_launch_attention(const omni_op_t* op, hipStream_t s) {
  switch (op->kind) {
    case OMNI_OP_ATTN_ROWS: return launch_attn_rows(op, s);
    case OMNI_OP_CHAN_ATTN: return launch_chan_attn(op, s);
    case OMNI_OP_ATTN_DECODE: return launch_attn_decode(op, s);
    default: omni_set_error("attention: bad kind %d", op->kind); return OMNI_E_ARG;
  }
}

int omni_launch_misc(const omni_op_t* op, hipStream_t s) {
  switch (op->kind) {
    case OMNI_OP_GREEDY_STEP: return launch_greedy(op, s);
    case OMNI_OP_CROP_RESIZE: return launch_crop_resize(op, s);
    case OMNI_OP_PROJ_PREP: case OMNI_OP_ASSEMBLE: case OMNI_OP_EMBED_STEP: return launch_glue(op, s);
    default: omni_set_error("misc: bad kind %d", op->kind); return OMNI_E_ARG;
  }
}
