// Pillow-exact separable resample (8 bits/channel, 22-bit fixed-point coefficients) fused with the
// letterbox paste and the /255 float conversion of the detector input.
//
// Replaces, on device, ref:util/yolov9.py:73-87 (_preprocess):
//   image.resize((rw, rh), LANCZOS) -> paste on a (114,114,114) canvas at (pad_left, pad_top)
//   -> np.float32 / 255.0 -> CHW.  (We emit NHWC with channels padded to the vector width.)
// Arithmetic follows Pillow's ImagingResampleHorizontal_8bpc / Vertical_8bpc: horizontal pass
// first, u8 intermediate, ss = (1 << 21) + sum(pixel * k), clip8(ss >> 22).  The coefficient
// tables come from omni_resample_coeffs() (host, double precision, same formulas as Pillow's
// precompute_coeffs + normalize_coeffs_8bpc).
#include "omni_internal.h"

#pragma clang fp contract(off)

namespace {

struct LbArgs {
  const unsigned char* img; unsigned char* tmp; const int* xb; const int* xk; const int* yb; const int* yk;
  void* y;
  int H, W, Hr, Wr, kx, ky, TH, TW, pad_left, pad_top, ldo, need_h, need_v, batch;
};

__device__ __forceinline__ int clip8(int v) {
  v >>= 22;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void resample_h_kernel(LbArgs a) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.H * a.Wr) return;
  int yrow = idx / a.Wr, xx = idx - yrow * a.Wr;
  int xmin = a.xb[xx * 2], cnt = a.xb[xx * 2 + 1];
  const int* k = a.xk + (long long)xx * a.kx;
  const unsigned char* src = a.img + ((long long)yrow * a.W + xmin) * 3;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int x = 0; x < cnt; ++x) {
    int kk = k[x];
    s0 += src[x * 3 + 0] * kk;
    s1 += src[x * 3 + 1] * kk;
    s2 += src[x * 3 + 2] * kk;
  }
  unsigned char* d = a.tmp + (long long)idx * 3;
  d[0] = (unsigned char)clip8(s0); d[1] = (unsigned char)clip8(s1); d[2] = (unsigned char)clip8(s2);
}

template <typename T>
__global__ __launch_bounds__(256) void resample_v_letterbox_kernel(LbArgs a) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.TH * a.TW) return;
  int ty = idx / a.TW, tx = idx - ty * a.TW;
  int yy = ty - a.pad_top, xx = tx - a.pad_left;
  int p0 = 114, p1 = 114, p2 = 114;
  if (yy >= 0 && yy < a.Hr && xx >= 0 && xx < a.Wr) {
    const unsigned char* src = a.need_h ? a.tmp : a.img;   // width already Wr
    if (a.need_v) {
      int ymin = a.yb[yy * 2], cnt = a.yb[yy * 2 + 1];
      const int* k = a.yk + (long long)yy * a.ky;
      int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
      for (int y = 0; y < cnt; ++y) {
        const unsigned char* p = src + ((long long)(ymin + y) * a.Wr + xx) * 3;
        int kk = k[y];
        s0 += p[0] * kk; s1 += p[1] * kk; s2 += p[2] * kk;
      }
      p0 = clip8(s0); p1 = clip8(s1); p2 = clip8(s2);
    } else {
      const unsigned char* p = src + ((long long)yy * a.Wr + xx) * 3;
      p0 = p[0]; p1 = p[1]; p2 = p[2];
    }
  }
  T* out = reinterpret_cast<T*>(a.y) + ((long long)a.batch * a.TH * a.TW + idx) * a.ldo;
  out[0] = ElemTraits<T>::from_f32((float)p0 / 255.0f);
  out[1] = ElemTraits<T>::from_f32((float)p1 / 255.0f);
  out[2] = ElemTraits<T>::from_f32((float)p2 / 255.0f);
  for (int c = 3; c < a.ldo; ++c) out[c] = ElemTraits<T>::from_f32(0.0f);
}

}  // namespace

int omni_launch_letterbox(const omni_op_t* op, hipStream_t s) {
  LbArgs a{};
  a.img = (const unsigned char*)op->p[0]; a.tmp = (unsigned char*)op->p[1];
  a.xb = (const int*)op->p[2]; a.xk = (const int*)op->p[3];
  a.y = op->p[4];
  a.yb = (const int*)op->p[5]; a.yk = (const int*)op->p[6];
  a.H = op->i[0]; a.W = op->i[1]; a.Hr = op->i[2]; a.Wr = op->i[3]; a.kx = op->i[4]; a.ky = op->i[5];
  a.TH = op->i[6]; a.TW = op->i[7]; a.pad_left = op->i[8]; a.pad_top = op->i[9];
  a.need_h = op->i[10]; a.need_v = op->i[11]; a.batch = op->i[12]; a.ldo = op->i[13];
  OMNI_REQUIRE(a.img && a.y, "letterbox: null pointer");
  OMNI_REQUIRE(a.H > 0 && a.W > 0 && a.Hr > 0 && a.Wr > 0 && a.TH >= a.Hr + a.pad_top && a.TW >= a.Wr + a.pad_left,
               "letterbox: bad geometry");
  OMNI_REQUIRE(a.ldo >= 3, "letterbox: ldo < 3");
  OMNI_REQUIRE(!a.need_h || (a.tmp && a.xb && a.xk && a.kx > 0), "letterbox: horizontal tables missing");
  OMNI_REQUIRE(!a.need_v || (a.yb && a.yk && a.ky > 0), "letterbox: vertical tables missing");
  OMNI_REQUIRE(a.need_h || a.W == a.Wr, "letterbox: need_h == 0 requires W == Wr");
  OMNI_REQUIRE(a.need_v || a.H == a.Hr, "letterbox: need_v == 0 requires H == Hr");
  if (a.need_h) {
    int n = a.H * a.Wr;
    hipLaunchKernelGGL(resample_h_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a);
  }
  int n = a.TH * a.TW;
  if (op->dtype == OMNI_F32)
    hipLaunchKernelGGL(resample_v_letterbox_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, a);
  else if (op->dtype == OMNI_F16)
    hipLaunchKernelGGL(resample_v_letterbox_kernel<half_t>, dim3((n + 255) / 256), dim3(256), 0, s, a);
  else OMNI_REQUIRE(false, "letterbox: bad dtype");
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}
