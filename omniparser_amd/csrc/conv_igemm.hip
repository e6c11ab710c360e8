// Implicit-GEMM convolution / linear layer for gfx950 (CDNA4), NHWC activations.
//
//   Y[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] ) (+ residual[m, n])
//   m = (b, ho, wo)   k = (r, s, c)   A[m,k] = X[b, ho*stride-pad+r, wo*stride-pad+s, in_coff+c]
//
// Replaces the Conv2d+BN+SiLU stack inside the reference's TorchScript detector
// (ref:util/yolov9.py:121) and every nn.Linear / patch-embed Conv2d of the captioner
// (hf:models/florence2/modeling_florence2.py:131-168, hf:models/bart/modeling_bart.py:143-257).
//
// Design (MI355X first):
//   * wave64, 4 waves / workgroup in a 2x2 arrangement, each wave owns (BM/2)x(BN/2) of the
//     output tile as 32x32 MFMA accumulators.
//   * OMNI_F32: v_mfma_f32_32x32x2_f32 — exact f32 products/accumulation (parity mode, 157 TF/s roof).
//     OMNI_F16: v_mfma_f32_32x32x16_f16 — f16 operands, f32 accumulate (2.5 PF/s roof).
//   * K is walked in 64-byte slices (16 f32 / 32 f16).  Both operand tiles are K-contiguous in
//     HBM (NHWC activations, [Cout][K] weights) so every global load is a 16-byte vector; tiles
//     are staged through LDS with 80-byte padded rows, which makes the MFMA-fragment
//     ds_read_b128 (row = lane&31, 16-byte column = lane>>5) bank-conflict free.
//   * K order inside a slice is permuted for the f32 path (lane-half h supplies k = 4h..4h+3 of
//     each 8-wide group): A and W use the same permutation so the contraction is unchanged while
//     one ds_read_b128 feeds four MFMAs.
//   * two LDS stages, one barrier per K slice; the next slice's global loads are issued before the
//     MFMAs of the current one and written to the other stage after them.
//   * split-K (f32 partials in a caller-provided workspace + a fixed-order reduce kernel that applies
//     bias/act/residual) fills the 256 CUs when M*Cout is small (P4/P5 layers at batch 1).
//   * concat / chunk are zero-copy: in_coff/ldi and out_coff/ldo address channel slices.
#include "omni_internal.h"
#include "gemm_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

OMNI_RANGE_GUARD_TU()

namespace {

struct ConvArgs {
  const void* x; const void* w; const float* bias; const void* res; void* y;
  int B, H, W, Cin, ldi, in_coff, KH, KW, stride, pad, Ho, Wo;
  int Cout, ldo, out_coff, act, ldr, res_coff;
  int M, K, ktiles, cin_tiles;
  int splits, kt_per_split, mtiles, ntiles, xcd_order, xcd_n;
  float* ws;
  int* cnt;               // split-K arrival counters, one per output tile (in-launch combine, round 6), or nullptr = reduce launch
  float scale;
  int vec_px;             // conv_split_kernel, row-patch mode (OMNI_OP_CONV i25): 16-byte vector v of a K slice is input pixel wi + v
};


// Epilogue shared by the register-staged kernels below.  C/D layout of a 32x32 MFMA tile: col = lane & 31 (output channel),
// row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (output pixel).  Per tile: all 16 residual loads are issued first (one
// wait instead of sixteen), the activation is a compile-time choice (no per-element switch) and addresses advance by row
// strides instead of being rebuilt per element.
template <typename T, int ACT, typename GetAcc>
__device__ __forceinline__ void epilogue_tile(const ConvArgs& a, int mb, int n, bool nok, float bias, GetAcc get) {
  if (!nok) return;
  T* __restrict__ Yb = reinterpret_cast<T*>(a.y) + (long long)mb * a.ldo + a.out_coff + n;
  const T* __restrict__ Rb = a.res ? reinterpret_cast<const T*>(a.res) + (long long)mb * a.ldr + a.res_coff + n : nullptr;
  float r[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int dm = (e & 3) + 8 * (e >> 2);
    r[e] = (Rb && mb + dm < a.M) ? ElemTraits<T>::to_f32(Rb[dm * a.ldr]) : 0.0f;
  }
  const float scale = a.scale;
  float amax = 0.0f;             // range guard (omni_internal.h): an f32 output beyond the f16 range cannot be split by the next conv's loader
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int dm = (e & 3) + 8 * (e >> 2);
    float v = get(e) + bias;
    if (scale != 0.0f) v *= scale;
    if constexpr (ACT == OMNI_ACT_SILU) v = v / (1.0f + expf(-v));                                   // torch CPU: x / (1 + exp(-x))
    else if constexpr (ACT == OMNI_ACT_GELU) v = omni_gelu(v);                                     // exact erf GELU
    v += r[e];
    if (mb + dm < a.M) {
      Yb[dm * a.ldo] = ElemTraits<T>::from_f32(v);
      amax = __builtin_fmaxf(__builtin_fabsf(v), amax);
    }
  }
  if constexpr (sizeof(T) == 4) omni_report_range(amax);
}

template <typename T, int BM, int BN, int RB, bool ALIGNED, bool PW>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int V = ElemTraits<T>::kVec;   // elements per 16-byte vector
  constexpr int VPR = RB / 16;             // 16-byte vectors per K-slice row
  constexpr int BKE = VPR * V;             // elements per K slice
  constexpr int ROWB = RB + 16;            // padded LDS row (bytes): conflict-free ds_read_b128
  constexpr int RPP = 256 / VPR;           // tile rows staged per pass of the 256 threads
  constexpr int A_IT = BM / RPP;
  constexpr int B_IT = BN / RPP;
  constexpr int TM = BM / 64;              // 32x32 tiles per wave along M
  constexpr int TN = BN / 64;
  constexpr int STAGE = (BM + BN) * ROWB;

  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; speed only).  Each XCD
  // walks all N tiles of ONE M tile back to back, so the activation tile stays in that XCD's 4 MiB L2
  // and is fetched from HBM once instead of once per N tile; the (small) weight matrix is shared via
  // L2/Infinity Cache by everybody.
  int mt, nt;
  if (!tile_of_block(blockIdx.x, a.mtiles, a.ntiles, a.xcd_order, a.xcd_n, mt, nt)) return;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  const int vec = tid % VPR;
  const int r0 = tid / VPR;

  // split-K: this block owns K slices [kt0, kt1)
  const int kt0 = blockIdx.z * a.kt_per_split;
  const int kt1 = min(kt0 + a.kt_per_split, a.ktiles);

  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wt = reinterpret_cast<const T*>(a.w);

  long long a_base[A_IT];
  int a_hi0[A_IT], a_wi0[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    int m = m0 + r0 + it * RPP;
    a_ok[it] = m < a.M;
    int mm = a_ok[it] ? m : 0;
    int wo = mm % a.Wo;
    int t = mm / a.Wo;
    int ho = t % a.Ho;
    int b = t / a.Ho;
    a_hi0[it] = ho * a.stride - a.pad;
    a_wi0[it] = wo * a.stride - a.pad;
    a_base[it] = (long long)b * a.H * a.W * a.ldi + a.in_coff;
  }
  const T* b_ptr[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    int n = n0 + r0 + it * RPP;
    b_ok[it] = n < a.Cout;
    b_ptr[it] = Wt + (long long)(b_ok[it] ? n : 0) * a.K;
  }

  u32x4 ra[A_IT], rb[B_IT];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // K-slice walker state: (tap r, tap s, first channel) advance incrementally — no divisions in the loop
  int w_r = 0, w_s = 0, w_c = 0;
  if (ALIGNED && !PW) {
    int tap = kt0 / a.cin_tiles;
    w_c = (kt0 - tap * a.cin_tiles) * BKE;
    w_r = tap / a.KW;
    w_s = tap - w_r * a.KW;
  }
  const T* a_row[A_IT];
  if (PW) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int m = m0 + r0 + it * RPP;
      a_row[it] = X + (long long)(a_ok[it] ? m : 0) * a.ldi + a.in_coff + vec * V;
    }
  }

  auto load_tile = [&](int kt) {
    if (PW) {
      // pointwise (1x1, stride 1, no padding): A is a plain row-major [M, ldi] matrix
#pragma unroll
      for (int it = 0; it < A_IT; ++it)
        ra[it] = a_ok[it] ? *reinterpret_cast<const u32x4*>(a_row[it] + kt * BKE) : zero4;
#pragma unroll
      for (int it = 0; it < B_IT; ++it)
        rb[it] = b_ok[it] ? *reinterpret_cast<const u32x4*>(b_ptr[it] + kt * BKE + vec * V) : zero4;
      return;
    }
    int r, s, c;
    bool kok = true;
    if (ALIGNED) {
      r = w_r; s = w_s; c = w_c + vec * V;
      w_c += BKE;
      if (w_c >= a.Cin) { w_c = 0; if (++w_s == a.KW) { w_s = 0; ++w_r; } }
    } else {
      int k = kt * BKE + vec * V;
      kok = k < a.K;
      int kk = kok ? k : 0;
      int tap = kk / a.Cin;
      c = kk - tap * a.Cin;
      r = tap / a.KW;
      s = tap - r * a.KW;
    }
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int hi = a_hi0[it] + r;
      int wi = a_wi0[it] + s;
      bool ok = kok && a_ok[it] && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      if (ok) {
        const T* p = X + a_base[it] + ((long long)hi * a.W + wi) * a.ldi + c;
        ra[it] = *reinterpret_cast<const u32x4*>(p);
      } else {
        ra[it] = zero4;
      }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (kok && b_ok[it]) {
        rb[it] = *reinterpret_cast<const u32x4*>(b_ptr[it] + kt * BKE + vec * V);
      } else {
        rb[it] = zero4;
      }
    }
  };
  auto store_tile = [&](int stage) {
    unsigned char* sA = lds + stage * STAGE;
    unsigned char* sB = sA + BM * ROWB;
#pragma unroll
    for (int it = 0; it < A_IT; ++it)
      *reinterpret_cast<u32x4*>(sA + (r0 + it * RPP) * ROWB + vec * 16) = ra[it];
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      *reinterpret_cast<u32x4*>(sB + (r0 + it * RPP) * ROWB + vec * 16) = rb[it];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int a_rd = (wm * (BM / 2) + (lane & 31)) * ROWB + (lane >> 5) * 16;
  const int b_rd = BM * ROWB + (wn * (BN / 2) + (lane & 31)) * ROWB + (lane >> 5) * 16;

  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile(0);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    if (more) load_tile(kt + 1);          // global loads in flight under the MFMAs below
    const unsigned char* st = lds + cur * STAGE;
#pragma unroll
    for (int kk = 0; kk < RB / 32; ++kk) {
      u32x4 av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[i] = *reinterpret_cast<const u32x4*>(st + a_rd + i * 32 * ROWB + kk * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bv[j] = *reinterpret_cast<const u32x4*>(st + b_rd + j * 32 * ROWB + kk * 32);
      if constexpr (sizeof(T) == 4) {
        f32x4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = __builtin_bit_cast(f32x4, av[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = __builtin_bit_cast(f32x4, bv[j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                __builtin_bit_cast(f16x8, av[i]), __builtin_bit_cast(f16x8, bv[j]), acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_tile(cur ^ 1);        // other stage: last read two iterations ago (barrier below orders it)
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if (a.splits > 1) {
    // raw f32 partial sums -> workspace [split][M][Cout]; bias/act/residual happen in the reduce kernel
    float* __restrict__ P = a.ws + (long long)blockIdx.z * a.M * a.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      if (n >= a.Cout) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int mb = m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int m = mb + (e & 3) + 8 * (e >> 2);
          if (m < a.M) P[(long long)m * a.Cout + n] = acc[i][j][e];
        }
      }
    }
    return;
  }
  auto run = [&](auto tag) {
    constexpr int ACT = decltype(tag)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      const bool nok = n < a.Cout;
      const float bias = (nok && a.bias) ? a.bias[n] : 0.0f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        epilogue_tile<T, ACT>(a, m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5), n, nok, bias, [&](int e) { return acc[i][j][e]; });
    }
  };
  if (a.act == OMNI_ACT_SILU) run(std::integral_constant<int, OMNI_ACT_SILU>{});
  else if (a.act == OMNI_ACT_GELU) run(std::integral_constant<int, OMNI_ACT_GELU>{});
  else run(std::integral_constant<int, OMNI_ACT_NONE>{});
}

// split-K second pass: y = act(sum_z partial[z] + bias) (+ residual); fixed summation order.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs a) {
  long long total = (long long)a.M * a.Cout;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int n = (int)(idx % a.Cout);
  long long m = idx / a.Cout;
  float v = 0.0f;
  for (int z = 0; z < a.splits; ++z) v += a.ws[(long long)z * total + idx];
  v += a.bias ? a.bias[n] : 0.0f;
  if (a.scale != 0.0f) v *= a.scale;
  v = act_apply(v, a.act);
  if (a.res) v += ElemTraits<T>::to_f32(reinterpret_cast<const T*>(a.res)[m * a.ldr + a.res_coff + n]);
  if constexpr (sizeof(T) == 4) omni_report_range(__builtin_fabsf(v));
  reinterpret_cast<T*>(a.y)[m * a.ldo + a.out_coff + n] = ElemTraits<T>::from_f32(v);
}


// ---------------------------------------------------------------------------------------------------------
// Split-f16 path: f32-class accuracy at the f16 MFMA rate.
//   a = ah + al * 2^-11,  w = wh + wl * 2^-11   (ah, wh = RTNE f16; al, wl = f16 of the scaled remainder)
//   a.w  ~=  ah.wh  +  2^-11 (ah.wl + al.wh)      (dropped al.wl term: 2^-22 relative)
// Three v_mfma_f32_32x32x16_f16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each) per
// 32x32x16 block: 5.3x fewer matrix-pipe cycles.  Products of f16 pairs are exact in f32; both partial sums
// accumulate in f32 (two accumulator sets) and are combined in the epilogue, so the result carries ~22 mantissa
// bits per operand — measured error vs f64 is at or below a plain f32 GEMM's accumulation error.
// Activations stay f32 in HBM and are split on the fly while staging to LDS; weights are split once on the host
// and stored as [Cout][K/16][16 hi | 16 lo] halves, i.e. the same 128 bytes per 32-wide K slice as f32 weights.
__device__ __forceinline__ void split_f16x4(const u32x4& raw, uint2& hi, uint2& lo) {
  // hi = f16(a) (round-toward-zero packed convert: any rounding works, lo absorbs the remainder exactly),
  // lo = f16((a - hi) * 2^11).  |a| must stay below the f16 range (65504) — true for every tensor on this path
  // (the reference itself runs this model in f16 on GPUs).
  typedef __fp16 hv2 __attribute__((ext_vector_type(2)));
  f32x4 v = __builtin_bit_cast(f32x4, raw);
  hv2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
  hv2 h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
  float r0 = (v[0] - (float)h01[0]) * 2048.0f, r1 = (v[1] - (float)h01[1]) * 2048.0f;
  float r2 = (v[2] - (float)h23[0]) * 2048.0f, r3 = (v[3] - (float)h23[1]) * 2048.0f;
  hv2 l01 = __builtin_amdgcn_cvt_pkrtz(r0, r1);
  hv2 l23 = __builtin_amdgcn_cvt_pkrtz(r2, r3);
  hi.x = __builtin_bit_cast(unsigned, h01); hi.y = __builtin_bit_cast(unsigned, h23);
  lo.x = __builtin_bit_cast(unsigned, l01); lo.y = __builtin_bit_cast(unsigned, l23);
}

// Split-K WITHOUT the second launch (round 6; OMNI_OP_CONV i24 / p6).  Round 4 built this with `__threadfence()` in every block and
// measured it 4.5x slower than the reduce launch (profiles/r4_s2_candidates_ab.txt) — a device-scope fence is a write-back of the
// XCD's whole L2 plus an invalidate.  This is the fence-free publish of the CDNA4 guide ("in-launch split-K reduction", sc1 form):
//   every split writes its partial tile WRITE-THROUGH (relaxed agent-scope atomic stores = `global_store_dword ... sc1`: the bytes
//   leave the XCD's L2 without a flush) -> each wave waits for its own stores (`s_waitcnt vmcnt(0)`) -> block barrier -> ONE lane
//   draws a ticket from the tile's arrival counter (relaxed agent-scope fetch_add) -> the block that draws `splits - 1` is the
//   tile's reducer: it reads all partials with agent-scope (sc1) loads — they bypass its XCD's L2 lines — sums them in split order
//   (the reduce kernel's order: bit-identical results), applies the ordinary epilogue and resets the counter for the next launch.
// Correct for any placement of a tile's splits over XCDs / CUs; no fence, no spinning (nobody waits for anybody).  The counters
// are zero before the first launch (host) and zero again after every launch (the reducers).
#ifndef OMNI_AGENT_ST_F32
#define OMNI_AGENT_ST_F32(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OMNI_AGENT_LD_F32(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OMNI_AGENT_ADD_I32(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OMNI_AGENT_ST_I32(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

// the split-order sum of one 32x32 accumulator tile's 16 elements of this lane: the partials of 2 splits in flight at a time.  Written
// for a SMALL register footprint — the reducer must not raise the kernel's allocation (a first version with 64-bit offsets and four
// splits in flight took conv_split_kernel<64,64,4> from 97 to 218 registers, i.e. from 4 to 2 waves per SIMD for every launch):
// 32-bit element offsets (a split-K launch has M * Cout * splits * 4 B <= the 32 MiB workspace), one uniform base per split.
__device__ __forceinline__ void combine_partials(const ConvArgs& a, int mb, int n, bool nok, float* sum) {
#pragma unroll
  for (int e = 0; e < 16; ++e) sum[e] = 0.0f;
  if (!nok) return;
  const int total = a.M * a.Cout;
  // rows beyond M (ragged last tile) read the last valid row instead of branching around every load; the epilogue drops them
  int off[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = mb + (e & 3) + 8 * (e >> 2);
    off[e] = (m < a.M ? m : a.M - 1) * a.Cout + n;
  }
  const float* __restrict__ base = a.ws;
  int z = 0;
  for (; z + 2 <= a.splits; z += 2, base += 2 * (long long)total) {
    float v0[16], v1[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v0[e] = OMNI_AGENT_LD_F32(base + off[e]);
#pragma unroll
    for (int e = 0; e < 16; ++e) v1[e] = OMNI_AGENT_LD_F32(base + total + off[e]);
#pragma unroll
    for (int e = 0; e < 16; ++e) sum[e] = (sum[e] + v0[e]) + v1[e];
  }
  if (z < a.splits) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sum[e] += OMNI_AGENT_LD_F32(base + off[e]);
  }
}

// Row-patch mode (round 6, OMNI_OP_CONV i25 = 1): a k x k convolution over a FEW input channels (the captioner's first patch embedding:
// 7 x 7, stride 4 over 3 channels stored as 4) has no 32-channel K slices — it ran on the exact-f32 MFMA kernel at 81 TF/s.  In NHWC
// with ld = 4 the k taps of one kernel ROW are k * 4 contiguous floats, so the layer is also a k x 1 convolution over "32 channels" =
// 8 consecutive pixels (k <= 8) whose weights are zero beyond tap k - 1: K = 32 k instead of 4 k^2, on the f16 matrix pipe.  The only
// difference to an ordinary slice is the bounds check — a vector of the slice is a pixel of its own (left / right image border, and
// the zero-weight pixel behind the last tap must not be read beyond the row: 0 x NaN) — hence `vec_px`.
template <int BM, int BN, int NW, bool PW>
__global__ __launch_bounds__(NW * 64, 2) void conv_split_kernel(ConvArgs a) {
  // NW waves as 2 (M) x NW/2 (N)
  constexpr int RB = 128, ROWB = RB + 16, VPR = 8, RPP = NW * 8;
  constexpr int BKE = 32;                  // f32 elements of K per slice
  constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
  constexpr int WN = NW / 2;
  constexpr int TM = BM / 64, TN = BN / (32 * WN);
  constexpr int STAGE = (BM + BN) * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int mt, nt;
  if (!tile_of_block(blockIdx.x, a.mtiles, a.ntiles, a.xcd_order, a.xcd_n, mt, nt)) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int vec = tid % VPR, r0 = tid / VPR;
  const int kt0 = blockIdx.z * a.kt_per_split;
  const int kt1 = min(kt0 + a.kt_per_split, a.ktiles);

  const float* __restrict__ X = reinterpret_cast<const float*>(a.x);
  const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(a.w);   // 4 bytes per (n, k)

  long long a_base[A_IT];
  int a_hi0[A_IT], a_wi0[A_IT];
  bool a_ok[A_IT];
  const float* a_row[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    int m = m0 + r0 + it * RPP;
    a_ok[it] = m < a.M;
    int mm = a_ok[it] ? m : 0;
    int wo = mm % a.Wo;
    int t = mm / a.Wo;
    int ho = t % a.Ho;
    int b = t / a.Ho;
    a_hi0[it] = ho * a.stride - a.pad;
    a_wi0[it] = wo * a.stride - a.pad;
    a_base[it] = (long long)b * a.H * a.W * a.ldi + a.in_coff;
    a_row[it] = X + (long long)mm * a.ldi + a.in_coff + vec * 4;
  }
  const unsigned char* b_ptr[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    int n = n0 + r0 + it * RPP;
    b_ok[it] = n < a.Cout;
    b_ptr[it] = Wb + (long long)(b_ok[it] ? n : 0) * a.K * 4 + vec * 16;
  }
  int w_r = 0, w_s = 0, w_c = 0;
  if (!PW) {
    int tap = kt0 / a.cin_tiles;
    w_c = (kt0 - tap * a.cin_tiles) * BKE;
    w_r = tap / a.KW;
    w_s = tap - w_r * a.KW;
  }

  u32x4 ra0[A_IT], rb0[B_IT];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_tile = [&](int kt, u32x4* ra, u32x4* rb) {
    if (PW) {
#pragma unroll
      for (int it = 0; it < A_IT; ++it)
        ra[it] = a_ok[it] ? *reinterpret_cast<const u32x4*>(a_row[it] + kt * BKE) : zero4;
    } else {
      int r = w_r, s = w_s, c = w_c + vec * 4;
      w_c += BKE;
      if (w_c >= a.Cin) { w_c = 0; if (++w_s == a.KW) { w_s = 0; ++w_r; } }
      const int dpx = vec * a.vec_px;          // row-patch mode: this vector IS pixel wi + vec (its own bounds); 0 otherwise
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        int hi = a_hi0[it] + r, wi = a_wi0[it] + s;
        bool ok = a_ok[it] && hi >= 0 && hi < a.H && wi + dpx >= 0 && wi + dpx < a.W;
        ra[it] = ok ? *reinterpret_cast<const u32x4*>(X + a_base[it] + ((long long)hi * a.W + wi) * a.ldi + c) : zero4;
      }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      rb[it] = b_ok[it] ? *reinterpret_cast<const u32x4*>(b_ptr[it] + (long long)kt * RB) : zero4;
  };
  // LDS row = two 16-wide K blocks, each [16 hi halves | 16 lo halves]
  const int a_wr = (vec >> 2) * 64 + (vec & 3) * 8;
  auto store_tile = [&](int stage, const u32x4* ra, const u32x4* rb) {
    unsigned char* sA = lds + stage * STAGE;
    unsigned char* sB = sA + BM * ROWB;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      uint2 hi, lo;
      split_f16x4(ra[it], hi, lo);
      unsigned char* p = sA + (r0 + it * RPP) * ROWB + a_wr;
      *reinterpret_cast<uint2*>(p) = hi;
      *reinterpret_cast<uint2*>(p + 32) = lo;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      *reinterpret_cast<u32x4*>(sB + (r0 + it * RPP) * ROWB + vec * 16) = rb[it];
  };

  f32x16 accM[TM][TN], accC[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { accM[i][j][e] = 0.0f; accC[i][j][e] = 0.0f; }

  const int a_rd = (wm * (BM / 2) + (lane & 31)) * ROWB + (lane >> 5) * 16;
  const int b_rd = BM * ROWB + (wn * (BN / WN) + (lane & 31)) * ROWB + (lane >> 5) * 16;

  auto compute = [&](int stage) {
    const unsigned char* st = lds + stage * STAGE;
#pragma unroll
    for (int j16 = 0; j16 < 2; ++j16) {
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(st + a_rd + i * 32 * ROWB + j16 * 64);
        al[i] = *reinterpret_cast<const f16x8*>(st + a_rd + i * 32 * ROWB + j16 * 64 + 32);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(st + b_rd + j * 32 * ROWB + j16 * 64);
        bl[j] = *reinterpret_cast<const f16x8*>(st + b_rd + j * 32 * ROWB + j16 * 64 + 32);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], accM[i][j], 0, 0, 0);
          accC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accC[i][j], 0, 0, 0);
          accC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accC[i][j], 0, 0, 0);
        }
    }
  };

  if (kt0 < kt1) { load_tile(kt0, ra0, rb0); store_tile(0, ra0, rb0); }
  __syncthreads();
  int cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    if (more) load_tile(kt + 1, ra0, rb0);
    compute(cur);
    if (more) store_tile(cur ^ 1, ra0, rb0);
    __syncthreads();
    cur ^= 1;
  }

  const float inv = 1.0f / 2048.0f;
  if (a.splits > 1) {
    float* __restrict__ P = a.ws + (long long)blockIdx.z * a.M * a.Cout;
    const bool combine = a.cnt != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
      if (n >= a.Cout) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int mb = m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int m = mb + (e & 3) + 8 * (e >> 2);
          if (m < a.M) {
            const float v = accM[i][j][e] + accC[i][j][e] * inv;
            if (combine) OMNI_AGENT_ST_F32(&P[(long long)m * a.Cout + n], v);      // write-through: published without a fence
            else P[(long long)m * a.Cout + n] = v;
          }
        }
      }
    }
    if (!combine) return;                                  // the reduce launch sums the partials
    // ---- in-launch combine: publish, draw the tile's ticket, the last arriver reduces
    OMNI_WAIT_VMCNT(0);                                    // this wave's partial stores have left
    __syncthreads();                                       // ... and every other wave's of this block
    int* const flag = reinterpret_cast<int*>(lds);         // (the K loop ended with a barrier: the ring is free; ONE __shared__ object)
    if (tid == 0) *flag = OMNI_AGENT_ADD_I32(&a.cnt[mt * a.ntiles + nt], 1);
    __syncthreads();
    if (*flag != a.splits - 1) return;
    if (tid == 0) OMNI_AGENT_ST_I32(&a.cnt[mt * a.ntiles + nt], 0);               // every split has arrived: ready for the next launch
    auto fin = [&](auto tag) {
      constexpr int ACT = decltype(tag)::value;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
        const bool nok = n < a.Cout;
        const float bias = (nok && a.bias) ? a.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5);
          float sum[16];
          combine_partials(a, mb, n, nok, sum);
          epilogue_tile<float, ACT>(a, mb, n, nok, bias, [&](int e) { return sum[e]; });
        }
      }
    };
    if (a.act == OMNI_ACT_SILU) fin(std::integral_constant<int, OMNI_ACT_SILU>{});
    else if (a.act == OMNI_ACT_GELU) fin(std::integral_constant<int, OMNI_ACT_GELU>{});
    else fin(std::integral_constant<int, OMNI_ACT_NONE>{});
    return;
  }
  auto run = [&](auto tag) {
    constexpr int ACT = decltype(tag)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
      const bool nok = n < a.Cout;
      const float bias = (nok && a.bias) ? a.bias[n] : 0.0f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        epilogue_tile<float, ACT>(a, m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5), n, nok, bias,
                                  [&](int e) { return accM[i][j][e] + accC[i][j][e] * inv; });
    }
  };
  if (a.act == OMNI_ACT_SILU) run(std::integral_constant<int, OMNI_ACT_SILU>{});
  else if (a.act == OMNI_ACT_GELU) run(std::integral_constant<int, OMNI_ACT_GELU>{});
  else run(std::integral_constant<int, OMNI_ACT_NONE>{});
}

template <int BM, int BN>
void launch_split_cfg(ConvArgs& a, hipStream_t s) {
  a.mtiles = (a.M + BM - 1) / BM;
  a.ntiles = (a.Cout + BN - 1) / BN;
  a.xcd_order = (a.mtiles >= 64 && a.ntiles > 1) ? 1 : 0;
  a.xcd_n = a.xcd_order ? choose_xcd_n(a.ntiles, 4ll * a.Cout * a.K) : 1;      // 4 bytes per (n, k): hi | lo halves
  dim3 grid(tile_grid(a.mtiles, a.ntiles, a.xcd_order, a.xcd_n), 1, a.splits);
  const bool pw = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo;
  // 128x128: 8 waves (64x32 per wave, 4 waves/SIMD; measured 191-229 TF/s vs 172-209 for 4 waves of 64x64); smaller
  // tiles: 4 waves.  Round-1 variants that lost (two-slice register prefetch, weights straight to registers, 256x128,
  // 64-wide K slices, weight-only LDS-DMA) are recorded in DESIGN.md and profiles/r2_gemm_diag.md, not kept here.
  if constexpr (BM == 128 && BN == 128) {
    if (pw) hipLaunchKernelGGL((conv_split_kernel<BM, BN, 8, true>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_split_kernel<BM, BN, 8, false>), grid, dim3(512), 0, s, a);
  } else {
    if (pw) hipLaunchKernelGGL((conv_split_kernel<BM, BN, 4, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_split_kernel<BM, BN, 4, false>), grid, dim3(256), 0, s, a);
  }
  if (a.splits > 1 && !a.cnt) {
    long long total = (long long)a.M * a.Cout;
    hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  }
}

// force_tile / force_splits (OMNI_OP_CONV i22 / i23, 0 = the heuristic below): the per-shape choices of the committed tuning table
// (omniparser_amd/conv_tuning_gfx950.json, measured by tools/conv_autotune.py as serialized graph replays on the MI355X).  The heuristic
// aims at >= 512-768 workgroups; for the latency-bound layers of a batch-1 detector pass (M = 400 ... 6 400: a kernel costs 5-13 us
// whatever it computes, and every split-K conv is followed by a ~5.5 us reduce launch — 24 % of the pass in round 5's kernel trace)
// fewer, longer blocks without the reduce are often faster.  Any choice gives the same sums up to the order of the K partials.
int launch_split(ConvArgs& a, long long ws_bytes, hipStream_t s, int force_tile, int force_splits, int* cnt, int n_cnt) {
  a.cin_tiles = a.Cin / 32;
  a.ktiles = a.K / 32;
  // 128x128 at 2 waves/SIMD is the fastest split tile (measured 171-207 TF/s vs 128-165 for 128x64); the
  // matrix pipe needs far fewer waves than the f32 path, so >= 512 workgroups is enough before shrinking tiles
  int bn = a.Cout > 64 ? 128 : 64, bm = 128;
  auto blocks = [&](int m, int n) { return (long long)((a.M + m - 1) / m) * ((a.Cout + n - 1) / n); };
  if (blocks(bm, bn) < 512 && bn == 128) bn = 64;
  if (blocks(bm, bn) < 512) bm = 64;
  if (force_tile) {
    OMNI_REQUIRE(force_tile >= 1 && force_tile <= 3, "conv: bad tile code %d (1 = 64x64, 2 = 128x64, 3 = 128x128)", force_tile);
    bm = force_tile == 1 ? 64 : 128;
    bn = force_tile == 3 ? 128 : 64;
  }
  long long nb = blocks(bm, bn);
  a.splits = 1;
  const long long cap = a.ws ? ws_bytes / ((long long)a.M * a.Cout * 4) : 1;
  if (force_splits) {
    OMNI_REQUIRE(force_splits >= 1 && force_splits <= 64, "conv: bad split count %d", force_splits);
    a.splits = force_splits > a.ktiles ? a.ktiles : force_splits;
    if (a.splits > cap) a.splits = (int)(cap < 1 ? 1 : cap);
  } else if (nb < 512 && a.ws) {
    int want = (int)((768 + nb - 1) / nb);
    int maxs = a.ktiles / 4;
    if (maxs > 32) maxs = 32;
    if (maxs > cap) maxs = (int)cap;
    a.splits = want < maxs ? want : maxs;
    if (a.splits < 2) a.splits = 1;
  }
  a.kt_per_split = (a.ktiles + a.splits - 1) / a.splits;
  a.splits = (a.ktiles + a.kt_per_split - 1) / a.kt_per_split;
  // in-launch combine when the caller provided arrival counters for every output tile (else: the reduce launch)
  a.cnt = (a.splits > 1 && cnt && nb <= n_cnt) ? cnt : nullptr;
  if (bm == 128 && bn == 128) launch_split_cfg<128, 128>(a, s);
  else if (bm == 128 && bn == 64) launch_split_cfg<128, 64>(a, s);
  else launch_split_cfg<64, 64>(a, s);
  return OMNI_OK;
}

template <typename T, int BM, int BN, int RB>
void launch_cfg(ConvArgs& a, bool aligned, hipStream_t s) {
  a.mtiles = (a.M + BM - 1) / BM;
  a.ntiles = (a.Cout + BN - 1) / BN;
  a.xcd_order = (a.mtiles >= 64 && a.ntiles > 1) ? 1 : 0;
  a.xcd_n = 1;                                  // exact-f32 / f16 kernels: row-block mapping only
  dim3 grid(tile_grid(a.mtiles, a.ntiles, a.xcd_order, a.xcd_n), 1, a.splits);
  const bool pw = aligned && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo;
  if (pw)
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, RB, true, true>), grid, dim3(256), 0, s, a);
  else if (aligned)
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, RB, true, false>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, RB, false, false>), grid, dim3(256), 0, s, a);
  if (a.splits > 1) {
    long long total = (long long)a.M * a.Cout;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  }
}

struct ConvCfg { int bm, bn, rb, splits, kt_per_split, ktiles, cin_tiles; bool aligned; };

// Tile / split-K choice.  MI355X has 256 CUs x 4 SIMDs; the f32 MFMA path hides its LDS + global
// latency only with >= 3-4 waves per SIMD, i.e. >= ~1024 four-wave workgroups in flight, so small-M
// layers (P4/P5 at batch 1: M = 1600 / 400) are split along K.
template <typename T>
ConvCfg choose_cfg(const ConvArgs& a, long long ws_bytes) {
  constexpr int V = ElemTraits<T>::kVec;
  ConvCfg c;
  c.rb = (a.Cin % (8 * V) == 0) ? 128 : 64;
  int bke = (c.rb / 16) * V;
  c.aligned = (a.Cin % bke) == 0;
  c.cin_tiles = c.aligned ? a.Cin / bke : 1;
  c.ktiles = (a.K + bke - 1) / bke;
  c.bn = a.Cout > 64 ? 128 : 64;
  c.bm = 128;
  auto blocks = [&](int m, int n) { return (long long)((a.M + m - 1) / m) * ((a.Cout + n - 1) / n); };
  if (blocks(c.bm, c.bn) < 1024 && c.bn == 128) c.bn = 64;
  if (blocks(c.bm, c.bn) < 1024) c.bm = 64;
  // 128-row tiles: 64-byte K slices keep LDS at 40 KB -> 3 workgroups per CU (measured +8..25 % over 128-byte
  // slices at 2 per CU); 64x64 tiles keep 128-byte slices (fewer barriers, LDS is not the limiter there)
  if (c.bm == 128) {
    if (c.rb == 128) {
      c.rb = 64;
      bke = (c.rb / 16) * V;
      c.aligned = (a.Cin % bke) == 0;
      c.cin_tiles = c.aligned ? a.Cin / bke : 1;
      c.ktiles = (a.K + bke - 1) / bke;
    }
  }
  long long nb = blocks(c.bm, c.bn);
  c.splits = 1;
  if (nb < 768 && a.ws) {
    int want = (int)((1024 + nb - 1) / nb);
    int maxs = c.ktiles / 4;                 // keep >= 4 K slices per split
    if (maxs > 32) maxs = 32;
    long long cap = ws_bytes / ((long long)a.M * a.Cout * 4);
    if (maxs > cap) maxs = (int)cap;
    c.splits = want < maxs ? want : maxs;
    if (c.splits < 2) c.splits = 1;
  }
  c.kt_per_split = (c.ktiles + c.splits - 1) / c.splits;
  c.splits = (c.ktiles + c.kt_per_split - 1) / c.kt_per_split;   // no empty splits
  return c;
}

template <typename T>
void launch_typed(ConvArgs& a, long long ws_bytes, hipStream_t s) {
  ConvCfg c = choose_cfg<T>(a, ws_bytes);
  a.cin_tiles = c.cin_tiles; a.ktiles = c.ktiles; a.splits = c.splits; a.kt_per_split = c.kt_per_split;
  if (c.rb == 128) {
    if (c.bm == 128 && c.bn == 128) launch_cfg<T, 128, 128, 128>(a, c.aligned, s);
    else if (c.bm == 128 && c.bn == 64) launch_cfg<T, 128, 64, 128>(a, c.aligned, s);
    else launch_cfg<T, 64, 64, 128>(a, c.aligned, s);
  } else {
    if (c.bm == 128 && c.bn == 128) launch_cfg<T, 128, 128, 64>(a, c.aligned, s);
    else if (c.bm == 128 && c.bn == 64) launch_cfg<T, 128, 64, 64>(a, c.aligned, s);
    else launch_cfg<T, 64, 64, 64>(a, c.aligned, s);
  }
}

}  // namespace

int omni_launch_conv(const omni_op_t* op, hipStream_t s) {
  ConvArgs a{};
  a.x = op->p[0]; a.w = op->p[1]; a.bias = (const float*)op->p[2]; a.res = op->p[3]; a.y = op->p[4];
  a.B = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.Cin = op->i[3]; a.ldi = op->i[4]; a.in_coff = op->i[5];
  a.KH = op->i[6]; a.KW = op->i[7]; a.stride = op->i[8]; a.pad = op->i[9]; a.Ho = op->i[10]; a.Wo = op->i[11];
  a.Cout = op->i[12]; a.ldo = op->i[13]; a.out_coff = op->i[14]; a.act = op->i[15];
  a.ldr = op->i[16]; a.res_coff = op->i[17];
  a.scale = op->f[0];
  a.ws = (float*)op->p[5];
  const long long ws_bytes = a.ws ? (long long)op->i[19] * 1024 : 0;   // i19 = workspace size in KiB
  a.splits = 1; a.kt_per_split = 0;
  const int V = op->dtype == OMNI_F32 ? 4 : 8;
  OMNI_REQUIRE(op->dtype == OMNI_F32 || op->dtype == OMNI_F16, "conv: bad dtype %d", op->dtype);
  OMNI_REQUIRE(a.x && a.w && a.y, "conv: null pointer");
  OMNI_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.Cout > 0 && a.Ho > 0 && a.Wo > 0, "conv: bad shape");
  OMNI_REQUIRE(a.Cin % V == 0 && a.ldi % V == 0 && a.in_coff % V == 0,
               "conv: Cin/ldi/in_coff (%d/%d/%d) must be multiples of %d", a.Cin, a.ldi, a.in_coff, V);
  OMNI_REQUIRE(a.KH > 0 && a.KW > 0 && a.stride > 0 && a.pad >= 0, "conv: bad window");
  OMNI_REQUIRE(a.act >= 0 && a.act <= OMNI_ACT_GELU, "conv: bad act %d", a.act);
  long long M = (long long)a.B * a.Ho * a.Wo;
  OMNI_REQUIRE(M < (1ll << 31), "conv: M too large");
  a.M = (int)M;
  a.K = a.KH * a.KW * a.Cin;
  a.vec_px = 0;
  if (op->i[25]) {        // row-patch mode: see conv_split_kernel
    OMNI_REQUIRE(op->i[25] == 1 && op->i[20] == 1 && a.KW == 1 && a.Cin == 32 && a.ldi == 4 && a.in_coff == 0 && a.KH <= 8,
                 "conv: row-patch mode (i25) needs i20 = 1, KW = 1, Cin = 32 (8 pixels of 4 channels), ldi = 4, in_coff = 0, KH <= 8");
    a.vec_px = 1;
  }
  if (op->i[20]) {        // split-f16 weights ([Cout][K/16][16 hi | 16 lo]) + f32 activations
    OMNI_REQUIRE(op->dtype == OMNI_F32 && a.Cin % 32 == 0, "conv: split-f16 mode needs f32 activations and Cin %% 32 == 0");
    OMNI_REQUIRE(op->i[24] >= 0 && (op->i[24] == 0 || op->p[6]), "conv: i24 arrival counters without p6");
    int rc = launch_split(a, ws_bytes, s, op->i[22], op->i[23], op->i[24] > 0 ? (int*)op->p[6] : nullptr, op->i[24]);
    if (rc) return rc;
  } else if (op->dtype == OMNI_F32) launch_typed<float>(a, ws_bytes, s);
  else launch_typed<half_t>(a, ws_bytes, s);
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

// Host mirror of the block -> tile mapping (same inline function the kernels use) so the permutation can be checked
// exhaustively without a GPU.  weight_bytes < 0: take xcd_n as given; otherwise it is chosen like the launcher does.
extern "C" int omni_debug_tile_map(int mtiles, int ntiles, int xcd_n, long long weight_bytes, int bid, int* mt, int* nt, int* grid,
                                   int* xcd_n_used) {
  if (mtiles <= 0 || ntiles <= 0 || !mt || !nt) return OMNI_E_ARG;
  const int xcd_order = (mtiles >= 64 && ntiles > 1) ? 1 : 0;
  int xn = xcd_order ? (weight_bytes >= 0 ? choose_xcd_n(ntiles, weight_bytes) : xcd_n) : 1;
  if (xn != 1 && xn != 2 && xn != 4 && xn != 8) return OMNI_E_ARG;
  if (ntiles % xn) return OMNI_E_ARG;
  if (grid) *grid = (int)tile_grid(mtiles, ntiles, xcd_order, xn);
  if (xcd_n_used) *xcd_n_used = xn;
  int a = -1, b = -1;
  bool ok = tile_of_block(bid, mtiles, ntiles, xcd_order, xn, a, b);
  *mt = ok ? a : -1;
  *nt = ok ? b : -1;
  return OMNI_OK;
}
