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
//   * next slice is prefetched into registers while the current one is multiplied.
//   * concat / chunk are zero-copy: in_coff/ldi and out_coff/ldo address channel slices.
#include "omni_internal.h"

namespace {

struct ConvArgs {
  const void* x; const void* w; const float* bias; const void* res; void* y;
  int B, H, W, Cin, ldi, in_coff, KH, KW, stride, pad, Ho, Wo;
  int Cout, ldo, out_coff, act, ldr, res_coff;
  int M, K, ktiles, cin_tiles;
  float scale;
};

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == OMNI_ACT_SILU) {
    // torch CPU: x / (1 + exp(-x))
    return v / (1.0f + expf(-v));
  } else if (act == OMNI_ACT_GELU) {
    // exact erf GELU (hf ACT2FN["gelu"]): 0.5 * x * (1 + erf(x / sqrt(2)))
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  }
  return v;
}

template <typename T, int BM, int BN, bool ALIGNED>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int V = ElemTraits<T>::kVec;   // elements per 16-byte vector
  constexpr int BKE = 4 * V;               // elements per 64-byte K slice
  constexpr int ROWB = 80;                 // padded LDS row (bytes)
  constexpr int A_IT = BM / 64;
  constexpr int B_IT = BN / 64;
  constexpr int TM = BM / 64;              // 32x32 tiles per wave along M
  constexpr int TN = BN / 64;

  __shared__ __attribute__((aligned(16))) unsigned char lds[(BM + BN) * ROWB];
  unsigned char* ldsA = lds;
  unsigned char* ldsB = lds + BM * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int vec = tid & 3;
  const int r0 = tid >> 2;

  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wt = reinterpret_cast<const T*>(a.w);

  long long a_base[A_IT];
  int a_hi0[A_IT], a_wi0[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    int m = m0 + r0 + it * 64;
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
    int n = n0 + r0 + it * 64;
    b_ok[it] = n < a.Cout;
    b_ptr[it] = Wt + (long long)(b_ok[it] ? n : 0) * a.K;
  }

  u32x4 ra[A_IT], rb[B_IT];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto load_tile = [&](int kt) {
    int tap, c;
    bool kok = true;
    if (ALIGNED) {
      tap = kt / a.cin_tiles;
      c = (kt - tap * a.cin_tiles) * BKE + vec * V;
    } else {
      int k = kt * BKE + vec * V;
      kok = k < a.K;
      int kk = kok ? k : 0;
      tap = kk / a.Cin;
      c = kk - tap * a.Cin;
    }
    int r = tap / a.KW;
    int s = tap - r * a.KW;
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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  load_tile(0);
  const int a_rd = (wm * (BM / 2) + (lane & 31)) * ROWB + (lane >> 5) * 16;
  const int b_rd = (wn * (BN / 2) + (lane & 31)) * ROWB + (lane >> 5) * 16;

  for (int kt = 0; kt < a.ktiles; ++kt) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it)
      *reinterpret_cast<u32x4*>(ldsA + (r0 + it * 64) * ROWB + vec * 16) = ra[it];
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      *reinterpret_cast<u32x4*>(ldsB + (r0 + it * 64) * ROWB + vec * 16) = rb[it];
    __syncthreads();
    if (kt + 1 < a.ktiles) load_tile(kt + 1);

#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[i] = *reinterpret_cast<const u32x4*>(ldsA + a_rd + i * 32 * ROWB + kk * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bv[j] = *reinterpret_cast<const u32x4*>(ldsB + b_rd + j * 32 * ROWB + kk * 32);
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
    __syncthreads();
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const T* __restrict__ R = reinterpret_cast<const T*>(a.res);
  const float scale = a.scale;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
    bool nok = n < a.Cout;
    float bias = (nok && a.bias) ? a.bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int mb = m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int m = mb + (e & 3) + 8 * (e >> 2);
        if (nok && m < a.M) {
          float v = acc[i][j][e] + bias;
          if (scale != 0.0f) v *= scale;
          v = act_apply(v, a.act);
          if (R) v += ElemTraits<T>::to_f32(R[(long long)m * a.ldr + a.res_coff + n]);
          Y[(long long)m * a.ldo + a.out_coff + n] = ElemTraits<T>::from_f32(v);
        }
      }
    }
  }
}

template <typename T, int BM, int BN>
void launch_cfg(const ConvArgs& a, bool aligned, hipStream_t s) {
  dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN, 1);
  if (aligned)
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, true>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, false>), grid, dim3(256), 0, s, a);
}

template <typename T>
void launch_typed(ConvArgs& a, hipStream_t s) {
  constexpr int V = ElemTraits<T>::kVec;
  constexpr int BKE = 4 * V;
  bool aligned = (a.Cin % BKE) == 0;
  a.cin_tiles = aligned ? a.Cin / BKE : 1;
  a.ktiles = (a.K + BKE - 1) / BKE;
  // tile choice: biggest tile that still yields >= ~2 workgroups per CU (256 CUs)
  int bn = a.Cout > 64 ? 128 : 64;
  int bm = 128;
  auto blocks = [&](int m, int n) { return (long long)((a.M + m - 1) / m) * ((a.Cout + n - 1) / n); };
  if (blocks(bm, bn) < 512 && bn == 128) bn = 64;
  if (blocks(bm, bn) < 512) bm = 64;
  if (bm == 128 && bn == 128) launch_cfg<T, 128, 128>(a, aligned, s);
  else if (bm == 128 && bn == 64) launch_cfg<T, 128, 64>(a, aligned, s);
  else launch_cfg<T, 64, 64>(a, aligned, s);
}

}  // namespace

int omni_launch_conv(const omni_op_t* op, hipStream_t s) {
  ConvArgs a;
  a.x = op->p[0]; a.w = op->p[1]; a.bias = (const float*)op->p[2]; a.res = op->p[3]; a.y = op->p[4];
  a.B = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.Cin = op->i[3]; a.ldi = op->i[4]; a.in_coff = op->i[5];
  a.KH = op->i[6]; a.KW = op->i[7]; a.stride = op->i[8]; a.pad = op->i[9]; a.Ho = op->i[10]; a.Wo = op->i[11];
  a.Cout = op->i[12]; a.ldo = op->i[13]; a.out_coff = op->i[14]; a.act = op->i[15];
  a.ldr = op->i[16]; a.res_coff = op->i[17];
  a.scale = op->f[0];
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
  if (op->dtype == OMNI_F32) launch_typed<float>(a, s);
  else launch_typed<half_t>(a, s);
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}
