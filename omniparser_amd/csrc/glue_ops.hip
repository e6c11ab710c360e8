// Detect -> caption hand-off on the device: what ref:util/utils.py:432-453 + remove_overlap_new (:241-319)
// + the crop arithmetic of get_parsed_content_icon (:90-98) do with Python lists between the two model stages, as one
// workgroup per screenshot, so the crop rectangles reach OMNI_OP_CROP_RESIZE without leaving HBM.
//
// Arithmetic mirrors the reference's types step by step (the reference mixes three of them):
//   * box / [w, h, w, h] is an f32 tensor division (IEEE);                                      ref :432, :438
//   * `.tolist()` turns the ratios into Python floats: every comparison of remove_overlap_new is f64, with the reference's
//     operation order (IoU with +1e-6 in the union, the two containment ratios, strict >);      ref :250-273
//   * int_box_area truncates f64 products `int(x * w)`;                                         ref :411-415
//   * crop corners are `int(coord * side)` on an f32 TENSOR element: an f32 product.            ref :95-96
// List semantics (App. E of SURVEY.md): an icon is dropped when some OTHER icon overlaps it above the threshold and is
// smaller; a kept icon collects the text of every OCR box lying inside it UNTIL the first OCR box it lies inside of (which
// drops the icon); each collected OCR box removes the first still-present dict-equal OCR entry (so the n-th donation to an
// equality class removes its n-th member — order independent, computed with one counter per class); output order is the
// stable sort "content is not None first": surviving OCR entries, icons with OCR text, icons without.
// Outputs: element table (kind, source index), donor bit masks (the host builds the label strings), crop rectangles in
// caption order, counts.  Host twin used as the test oracle: omniparser_amd/pipeline.py::ScreenParser.glue.
#include "omni_internal.h"
#include <string.h>

namespace {

constexpr int GLUE_MAX_ICONS = 512;
constexpr int GLUE_MAX_OCR = 1024;

struct GlueArgs {
  const float* boxes; const int* count; const double* ocr; const int* ocr_meta;
  int* elems; int* crops; int* counts; unsigned long long* donors;
  int max_det, n_ocr, W, H, mw, boxes_are_ratio, cap_elems;
  double thr;
};

__device__ __forceinline__ double inter_area(const double* a, const double* b) {
  const double x1 = fmax(a[0], b[0]), y1 = fmax(a[1], b[1]), x2 = fmin(a[2], b[2]), y2 = fmin(a[3], b[3]);
  return fmax(0.0, x2 - x1) * fmax(0.0, y2 - y1);
}

// exclusive prefix sum of per-thread flags over items [0, n) in item order, 256 threads, items strided by thread
__device__ int block_scan_flags(const unsigned char* flag, int n, int* pos, int* scratch) {
  // chunked: thread t owns items [t*per, (t+1)*per)
  const int t = threadIdx.x, per = (n + 255) / 256;
  int c = 0;
  for (int i = t * per; i < min(n, (t + 1) * per); ++i) c += flag[i];
  scratch[t] = c;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int k = 0; k < 256; ++k) { int v = scratch[k]; scratch[k] = run; run += v; }
    scratch[256] = run;
  }
  __syncthreads();
  int run = scratch[t];
  for (int i = t * per; i < min(n, (t + 1) * per); ++i) { pos[i] = run; run += flag[i]; }
  const int total = scratch[256];
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(256) void glue_kernel(GlueArgs a) {
  __shared__ double ib[GLUE_MAX_ICONS][4];          // icon ratios as Python floats
  __shared__ float ibf[GLUE_MAX_ICONS][4];          // the same ratios as the f32 tensor elements they came from
  __shared__ double ia[GLUE_MAX_ICONS];
  __shared__ unsigned char live[GLUE_MAX_ICONS];    // survives int_box_area
  __shared__ unsigned char state[GLUE_MAX_ICONS];   // 0 dropped, 1 kept with OCR text, 2 kept without
  __shared__ unsigned char oalive[GLUE_MAX_OCR];
  __shared__ int ecount[GLUE_MAX_OCR];              // donation events per OCR equality class
  __shared__ int pos[GLUE_MAX_OCR > GLUE_MAX_ICONS ? GLUE_MAX_OCR : GLUE_MAX_ICONS];
  __shared__ unsigned char flag[GLUE_MAX_OCR > GLUE_MAX_ICONS ? GLUE_MAX_OCR : GLUE_MAX_ICONS];
  __shared__ int scratch[257];
  __shared__ int el_s[GLUE_MAX_ICONS + GLUE_MAX_OCR];   // element table (kind << 16 | source) kept in LDS for the crop phase
  const int t = threadIdx.x;
  const int K = min(*a.count, a.max_det);
  const int M = a.n_ocr > 0 ? min(a.ocr_meta[0], a.n_ocr) : 0;      // live OCR count: first word of the meta table (plans are static)
  const int* meta = a.ocr_meta + 2;
  const float fw = (float)a.W, fh = (float)a.H;
  const double dw = (double)a.W, dh = (double)a.H;

  for (int i = t; i < K; i += 256) {
    float r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v = a.boxes[i * 4 + c];
      r[c] = a.boxes_are_ratio ? v : v / ((c & 1) ? fh : fw);          // f32 tensor division (ref :432)
      ibf[i][c] = r[c];
      ib[i][c] = (double)r[c];                                          // .tolist()
    }
    const int x1 = (int)(ib[i][0] * dw), y1 = (int)(ib[i][1] * dh), x2 = (int)(ib[i][2] * dw), y2 = (int)(ib[i][3] * dh);
    live[i] = ((x2 - x1) * (y2 - y1)) > 0;                              // int_box_area > 0 (ref :445)
    ia[i] = (ib[i][2] - ib[i][0]) * (ib[i][3] - ib[i][1]);
    state[i] = 0;
  }
  for (int j = t; j < M; j += 256) { ecount[j] = 0; oalive[j] = 1; }
  __syncthreads();

  // ---- icon vs icon: drop the larger box of an overlapping pair (compared against ALL live icons, dropped or not)
  for (int i = t; i < K; i += 256) {
    if (!live[i]) continue;
    bool valid = true;
    const double a1 = ia[i];
    for (int j = 0; j < K && valid; ++j) {
      if (j == i || !live[j]) continue;
      const double it = inter_area(ib[i], ib[j]);
      const double a2 = ia[j];
      const double uni = a1 + a2 - it + 1e-6;
      double r1 = 0.0, r2 = 0.0;
      if (a1 > 0 && a2 > 0) { r1 = it / a1; r2 = it / a2; }
      const double ov = fmax(fmax(it / uni, r1), r2);
      if (ov > a.thr && a1 > a2) valid = false;
    }
    if (!valid) continue;
    // ---- icon vs OCR, in OCR order
    unsigned long long* dm = a.donors + (size_t)i * a.mw;
    for (int wd = 0; wd < a.mw; ++wd) dm[wd] = 0ull;
    bool swallowed = false, any = false;
    for (int j = 0; j < M; ++j) {
      const double* ob = a.ocr + (size_t)j * 4;
      const double it = inter_area(ob, ib[i]);
      const double oa = (ob[2] - ob[0]) * (ob[3] - ob[1]);
      if (it / oa > 0.80) {                                             // OCR box inside the icon: donate its text
        dm[j >> 6] |= 1ull << (j & 63);
        any = true;
        atomicAdd(&ecount[meta[2 * j]], 1);
      } else if (it / a1 > 0.80) {                                      // icon inside an OCR box: drop the icon, stop scanning
        swallowed = true;
        break;
      }
    }
    if (!swallowed) state[i] = any ? 1 : 2;
  }
  __syncthreads();
  // `list.remove(elem)` removes the first remaining equal entry: the n-th donation of a class removes its n-th member
  for (int j = t; j < M; j += 256) oalive[j] = meta[2 * j + 1] >= ecount[meta[2 * j]];
  __syncthreads();

  // ---- ordered element table: surviving OCR | icons with text | icons without (stable sort on `content is None`)
  for (int j = t; j < M; j += 256) flag[j] = oalive[j];
  __syncthreads();
  const int n_ocr = block_scan_flags(flag, M, pos, scratch);
  for (int j = t; j < M; j += 256)
    if (oalive[j] && pos[j] < a.cap_elems) { a.elems[2 * pos[j]] = 0; a.elems[2 * pos[j] + 1] = j; el_s[pos[j]] = j; }
  __syncthreads();
  for (int i = t; i < K; i += 256) flag[i] = state[i] == 1;
  __syncthreads();
  const int n_lab = block_scan_flags(flag, K, pos, scratch);
  for (int i = t; i < K; i += 256)
    if (state[i] == 1 && n_ocr + pos[i] < a.cap_elems) { a.elems[2 * (n_ocr + pos[i])] = 1; a.elems[2 * (n_ocr + pos[i]) + 1] = i; el_s[n_ocr + pos[i]] = (1 << 16) | i; }
  __syncthreads();
  for (int i = t; i < K; i += 256) flag[i] = state[i] == 2;
  __syncthreads();
  const int n_none = block_scan_flags(flag, K, pos, scratch);
  const int base = n_ocr + n_lab;
  for (int i = t; i < K; i += 256)
    if (state[i] == 2 && base + pos[i] < a.cap_elems) { a.elems[2 * (base + pos[i])] = 2; a.elems[2 * (base + pos[i]) + 1] = i; el_s[base + pos[i]] = (2 << 16) | i; }
  const int n_el = base + n_none;
  const int start = n_none > 0 ? base : -1;                             // index of the first `content is None` (ref :451)
  __syncthreads();

  // ---- crop rectangles of `filtered_boxes[start:]` (all boxes when start == 0, the LAST box when start == -1: ref :90-93)
  const int first = start > 0 ? start : (start == 0 ? 0 : max(n_el - 1, 0));
  const int n_src = n_el == 0 ? 0 : n_el - first;
  // candidates are elements first..n_el-1; when start > 0 they are exactly the icons of state 2 in order
  for (int q = t; q < GLUE_MAX_ICONS; q += 256) flag[q] = 0;
  __syncthreads();
  __shared__ int rect[GLUE_MAX_ICONS][4];
  for (int q = t; q < n_src && q < GLUE_MAX_ICONS; q += 256) {
    const int e = first + q;
    const int kind = el_s[e] >> 16, src = el_s[e] & 0xffff;
    float r[4];
    if (kind == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = (float)a.ocr[(size_t)src * 4 + c];    // torch.tensor(list of floats) -> f32 (exact: they came from f32)
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = ibf[src][c];
    }
    const int x0 = (int)(r[0] * fw), x1 = (int)(r[2] * fw), y0 = (int)(r[1] * fh), y1 = (int)(r[3] * fh);   // f32 products (ref :95-96)
    const bool ok = !(x1 - x0 <= 0 || y1 - y0 <= 0 || x0 < 0 || y0 < 0);
    flag[q] = ok;
    rect[q][0] = x0; rect[q][1] = y0; rect[q][2] = min(x1, a.W); rect[q][3] = min(y1, a.H);
  }
  __syncthreads();
  const int n_crop = block_scan_flags(flag, min(n_src, GLUE_MAX_ICONS), pos, scratch);
  for (int q = t; q < n_src && q < GLUE_MAX_ICONS; q += 256)
    if (flag[q]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) a.crops[4 * pos[q] + c] = rect[q][c];
    }
  if (t == 0) { a.counts[0] = n_el; a.counts[1] = n_crop; a.counts[2] = start; a.counts[3] = n_ocr; }
}

}  // namespace

// argument unpacking shared by the launcher and the host emulation
static int glue_args_from_op(const omni_op_t* op, GlueArgs& a, const char** why) {
  a.boxes = (const float*)op->p[0]; a.count = (const int*)op->p[1]; a.ocr = (const double*)op->p[2]; a.ocr_meta = (const int*)op->p[3];
  a.elems = (int*)op->p[4]; a.crops = (int*)op->p[5]; a.counts = (int*)op->p[6]; a.donors = (unsigned long long*)op->p[7];
  a.max_det = op->i[0]; a.n_ocr = op->i[1]; a.W = op->i[2]; a.H = op->i[3]; a.mw = op->i[4]; a.boxes_are_ratio = op->i[5]; a.cap_elems = op->i[6];
  a.thr = (double)op->f[0];
  if (op->i[7]) {                       // threshold as an exact double (two i32 halves): Python passes 0.7 / 0.9 as f64
    unsigned long long bits = ((unsigned long long)(unsigned)op->i[9] << 32) | (unsigned)op->i[8];
    memcpy(&a.thr, &bits, 8);
  }
  *why = nullptr;
  if (!(a.boxes && a.count && a.elems && a.crops && a.counts && a.donors)) *why = "glue: null pointer";
  else if (!(a.max_det > 0 && a.max_det <= GLUE_MAX_ICONS && a.n_ocr >= 0 && a.n_ocr <= GLUE_MAX_OCR && a.W > 0 && a.H > 0))
    *why = "glue: capacity (icons <= 512, OCR boxes <= 1024)";
  else if (!(a.n_ocr == 0 || (a.ocr && a.ocr_meta))) *why = "glue: OCR boxes without their tables";
  else if (!(a.mw >= (a.n_ocr + 63) / 64 && a.mw >= 1 && a.cap_elems >= a.max_det + a.n_ocr)) *why = "glue: output capacity";
  return *why ? 1 : 0;
}

// OMNI_OP_GLUE (see include/omni_amd.h)
int omni_launch_glue(const omni_op_t* op, hipStream_t s) {
  GlueArgs a{};
  const char* why = nullptr;
  if (glue_args_from_op(op, a, &why)) { omni_set_error("%s", why); return OMNI_E_ARG; }
  hipLaunchKernelGGL(glue_kernel, dim3(1), dim3(256), 0, s, a);
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}
