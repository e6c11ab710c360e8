// HBM-bound spatial ops of the YOLOv9-E graph (NHWC, 16-byte vectors per lane):
//   avg_pool2d(2,1,0) and max_pool2d(3,2,1) of ADown, max_pool2d(5,1,2) of SPPELAN,
//   F.interpolate(mode='nearest') of nn.Upsample / CBFuse (with optional accumulate).
// They replace ATen pooling / upsample kernels reached through the TorchScript blob
// (ref:util/yolov9.py:121).  Each lane moves one 16-byte channel vector; consecutive lanes
// walk consecutive channels of one pixel, so a wave reads/writes contiguous 1 KiB segments.
#include "omni_internal.h"

namespace {

struct PoolArgs {
  const void* x; void* y;
  int B, H, W, C, ldi, in_coff, k, stride, pad, Ho, Wo, ldo, out_coff, accumulate;
  float hscale, wscale;
  long long total;  // B*Ho*Wo*(C/V)
  // MODE 3 (CBFuse): up to 5 sources of one channel count, each with its own size / pitch / slice; source 0 = x, H, W, ldi, in_coff
  int nsrc;
  const void* xs[4]; int Hs[4], Ws[4], lds[4], coffs[4];
};

template <typename T> struct Vec16 {
  static constexpr int V = ElemTraits<T>::kVec;
  T v[V];
};

template <typename T>
__device__ __forceinline__ Vec16<T> ldv(const T* p) {
  u32x4 r = *reinterpret_cast<const u32x4*>(p);
  return __builtin_bit_cast(Vec16<T>, r);
}
template <typename T>
__device__ __forceinline__ void stv(T* p, const Vec16<T>& v) {
  *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, v);
}

// ATen nearest_idx: same size -> identity; exact 2x -> >>1; else min(floor(dst * scale), in - 1) with scale = in / out in f32
__device__ __forceinline__ int nearest_idx(int o, int out, int in) {
  if (out == in) return o;
  if (out == 2 * in) return o >> 1;
  int i = (int)floorf(o * ((float)in / (float)out));
  return i < in - 1 ? i : in - 1;
}

template <typename T, int MODE>  // MODE 0 avgpool2 s1, 1 maxpool, 2 nearest resize, 3 sum of nearest-resized sources (CBFuse)
__global__ __launch_bounds__(256) void pool_kernel(PoolArgs a) {
  constexpr int V = ElemTraits<T>::kVec;
  const int cv = a.C / V;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < a.total;
       idx += (long long)gridDim.x * blockDim.x) {
    int c = (int)(idx % cv) * V;
    long long pix = idx / cv;
    int wo = (int)(pix % a.Wo);
    long long t = pix / a.Wo;
    int ho = (int)(t % a.Ho);
    int b = (int)(t / a.Ho);
    const T* xb = X + (long long)b * a.H * a.W * a.ldi + a.in_coff + c;
    T* yp = Y + pix * a.ldo + a.out_coff + c;
    float acc[V];
    if (MODE == 0) {
      // torch cpu_avg_pool2d: sum = 0; for ih: for iw: sum += x; out = sum / 4
      Vec16<T> v00 = ldv<T>(xb + ((long long)ho * a.W + wo) * a.ldi);
      Vec16<T> v01 = ldv<T>(xb + ((long long)ho * a.W + wo + 1) * a.ldi);
      Vec16<T> v10 = ldv<T>(xb + ((long long)(ho + 1) * a.W + wo) * a.ldi);
      Vec16<T> v11 = ldv<T>(xb + ((long long)(ho + 1) * a.W + wo + 1) * a.ldi);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float s = ElemTraits<T>::to_f32(v00.v[e]);
        s += ElemTraits<T>::to_f32(v01.v[e]);
        s += ElemTraits<T>::to_f32(v10.v[e]);
        s += ElemTraits<T>::to_f32(v11.v[e]);
        acc[e] = s * 0.25f;
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = -INFINITY;
      int h0 = ho * a.stride - a.pad, w0 = wo * a.stride - a.pad;
      for (int r = 0; r < a.k; ++r) {
        int hi = h0 + r;
        if (hi < 0 || hi >= a.H) continue;
        for (int s = 0; s < a.k; ++s) {
          int wi = w0 + s;
          if (wi < 0 || wi >= a.W) continue;
          Vec16<T> v = ldv<T>(xb + ((long long)hi * a.W + wi) * a.ldi);
#pragma unroll
          for (int e = 0; e < V; ++e) acc[e] = fmaxf(acc[e], ElemTraits<T>::to_f32(v.v[e]));
        }
      }
    } else if (MODE == 3) {
      // CBFuse (ref blob: torch.sum(torch.stack([F.interpolate(x_i, size, 'nearest') ...]), 0)): ((s0 + s1) + s2) + ... in source order,
      // every partial sum rounded to T — bit for bit what the chain of accumulate-resize launches computed, in ONE pass over the output
      Vec16<T> v = ldv<T>(xb + ((long long)nearest_idx(ho, a.Ho, a.H) * a.W + nearest_idx(wo, a.Wo, a.W)) * a.ldi);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = ElemTraits<T>::to_f32(v.v[e]);
      for (int k = 0; k + 1 < a.nsrc; ++k) {
        const T* xk = reinterpret_cast<const T*>(a.xs[k]) + (long long)b * a.Hs[k] * a.Ws[k] * a.lds[k] + a.coffs[k] + c;
        Vec16<T> u = ldv<T>(xk + ((long long)nearest_idx(ho, a.Ho, a.Hs[k]) * a.Ws[k] + nearest_idx(wo, a.Wo, a.Ws[k])) * a.lds[k]);
#pragma unroll
        for (int e = 0; e < V; ++e)
          acc[e] = ElemTraits<T>::to_f32(ElemTraits<T>::from_f32(acc[e])) + ElemTraits<T>::to_f32(u.v[e]);
      }
    } else {
      // ATen nearest_idx: same size -> identity; exact 2x -> >>1; else min(floor(dst*scale), in-1)
      int hi, wi;
      if (a.Ho == a.H) hi = ho; else if (a.Ho == 2 * a.H) hi = ho >> 1;
      else { hi = (int)floorf(ho * a.hscale); hi = hi < a.H - 1 ? hi : a.H - 1; }
      if (a.Wo == a.W) wi = wo; else if (a.Wo == 2 * a.W) wi = wo >> 1;
      else { wi = (int)floorf(wo * a.wscale); wi = wi < a.W - 1 ? wi : a.W - 1; }
      Vec16<T> v = ldv<T>(xb + ((long long)hi * a.W + wi) * a.ldi);
      if (a.accumulate) {
        Vec16<T> o = ldv<T>(yp);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = ElemTraits<T>::to_f32(o.v[e]) + ElemTraits<T>::to_f32(v.v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = ElemTraits<T>::to_f32(v.v[e]);
      }
    }
    Vec16<T> out;
#pragma unroll
    for (int e = 0; e < V; ++e) out.v[e] = ElemTraits<T>::from_f32(acc[e]);
    stv<T>(yp, out);
  }
}

int fill_args(const omni_op_t* op, PoolArgs& a, const char* name) {
  a.x = op->p[0]; a.y = op->p[4];
  a.B = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.C = op->i[3]; a.ldi = op->i[4]; a.in_coff = op->i[5];
  a.k = op->i[6]; a.stride = op->i[8]; a.pad = op->i[9]; a.Ho = op->i[10]; a.Wo = op->i[11];
  a.ldo = op->i[13]; a.out_coff = op->i[14]; a.accumulate = op->i[18];
  const int V = op->dtype == OMNI_F32 ? 4 : 8;
  OMNI_REQUIRE(op->dtype == OMNI_F32 || op->dtype == OMNI_F16, "%s: bad dtype", name);
  OMNI_REQUIRE(a.x && a.y, "%s: null pointer", name);
  OMNI_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0, "%s: bad shape", name);
  OMNI_REQUIRE(a.C % V == 0 && a.ldi % V == 0 && a.in_coff % V == 0 && a.ldo % V == 0 && a.out_coff % V == 0,
               "%s: channel counts/offsets must be multiples of %d", name, V);
  return OMNI_OK;
}

template <int MODE>
int launch(const omni_op_t* op, PoolArgs& a, hipStream_t s) {
  const int V = op->dtype == OMNI_F32 ? 4 : 8;
  a.total = (long long)a.B * a.Ho * a.Wo * (a.C / V);
  long long blocks = (a.total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (op->dtype == OMNI_F32)
    hipLaunchKernelGGL((pool_kernel<float, MODE>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((pool_kernel<half_t, MODE>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

}  // namespace

int omni_launch_avgpool2(const omni_op_t* op, hipStream_t s) {
  PoolArgs a{};
  int rc = fill_args(op, a, "avgpool2");
  if (rc) return rc;
  OMNI_REQUIRE(a.H > 1 && a.W > 1, "avgpool2: input too small");
  a.Ho = a.H - 1; a.Wo = a.W - 1;
  return launch<0>(op, a, s);
}

int omni_launch_maxpool(const omni_op_t* op, hipStream_t s) {
  PoolArgs a{};
  int rc = fill_args(op, a, "maxpool");
  if (rc) return rc;
  OMNI_REQUIRE(a.k > 0 && a.stride > 0 && a.pad >= 0 && a.pad * 2 <= a.k, "maxpool: bad window");
  OMNI_REQUIRE(a.Ho == (a.H + 2 * a.pad - a.k) / a.stride + 1 && a.Wo == (a.W + 2 * a.pad - a.k) / a.stride + 1,
               "maxpool: Ho/Wo mismatch");
  return launch<1>(op, a, s);
}

int omni_launch_resize_nearest(const omni_op_t* op, hipStream_t s) {
  PoolArgs a{};
  int rc = fill_args(op, a, "resize_nearest");
  if (rc) return rc;
  OMNI_REQUIRE(a.Ho > 0 && a.Wo > 0, "resize_nearest: bad output size");
  if (op->i[17] > 1) {                       // CBFuse: i17 sources summed in order (include/omni_amd.h)
    a.nsrc = op->i[17];
    OMNI_REQUIRE(a.nsrc <= 5 && !a.accumulate, "resize_nearest: at most 5 sources, and no accumulate with several");
    static const int pidx[4] = {1, 2, 3, 5};
    static const int base[4][4] = {{19, 20, 21, 22}, {23, 24, 25, 26}, {27, 28, 29, 30}, {7, 12, 15, 16}};
    const int V = op->dtype == OMNI_F32 ? 4 : 8;
    for (int k = 0; k + 1 < a.nsrc; ++k) {
      a.xs[k] = op->p[pidx[k]];
      a.Hs[k] = op->i[base[k][0]]; a.Ws[k] = op->i[base[k][1]]; a.lds[k] = op->i[base[k][2]]; a.coffs[k] = op->i[base[k][3]];
      OMNI_REQUIRE(a.xs[k] && a.Hs[k] > 0 && a.Ws[k] > 0 && a.lds[k] % V == 0 && a.coffs[k] % V == 0, "resize_nearest: bad source %d", k + 1);
    }
    return launch<3>(op, a, s);
  }
  a.hscale = (float)a.H / (float)a.Ho;
  a.wscale = (float)a.W / (float)a.Wo;
  return launch<2>(op, a, s);
}
