// C ABI of libomni_amd.so: error channel, single-op dispatch, the plan executor
// (eager replay or hipGraph replay of an immutable op list) and the host-side
// Pillow coefficient generator.  See include/omni_amd.h for the contract.
#include "omni_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

static thread_local char g_err[512] = "";

void omni_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* omni_last_error(void) { return g_err; }
extern "C" int omni_abi_version(void) { return OMNI_ABI_VERSION; }

// ---- range guard of the split formats (omni_internal.h): one device counter per translation unit, read through registered readers
static omni_ovf_reader_t g_ovf_readers[8];
static int g_ovf_n = 0;
void omni_register_overflow_reader(omni_ovf_reader_t fn) {
  if (g_ovf_n < 8) g_ovf_readers[g_ovf_n++] = fn;
}
extern "C" int omni_overflow_count(int reset, unsigned long long* count) {
  OMNI_REQUIRE(count != nullptr, "omni_overflow_count: null result pointer");
  unsigned long long total = 0;
  for (int k = 0; k < g_ovf_n; ++k) {
    unsigned int v = 0;
    if (g_ovf_readers[k](&v, reset)) {
      omni_set_error("omni_overflow_count: reading the device counter failed");
      return OMNI_E_HIP;
    }
    total += v;
  }
  *count = total;
  return OMNI_OK;
}

extern "C" int omni_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    omni_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return OMNI_E_NODEV;
  }
  return n;
}

static int dispatch(const omni_op_t* op, hipStream_t s) {
  switch (op->kind) {
    case OMNI_OP_CONV: return op->i[20] == 2 ? omni_launch_gemm_dma(op, s) : omni_launch_conv(op, s);
    case OMNI_OP_SPLIT_CONVERT: return omni_launch_split_convert(op, s);
    case OMNI_OP_MLP_FUSED: return omni_launch_mlp_fused(op, s);
    case OMNI_OP_GLUE: return omni_launch_glue(op, s);
    case OMNI_OP_OVERLAY: return omni_launch_overlay(op, s);
    case OMNI_OP_PNG_PACK: return omni_launch_png_pack(op, s);
    case OMNI_OP_PNG_DEFLATE: return omni_launch_png_deflate(op, s);
    case OMNI_OP_AVGPOOL2: return omni_launch_avgpool2(op, s);
    case OMNI_OP_MAXPOOL: return omni_launch_maxpool(op, s);
    case OMNI_OP_RESIZE_NEAREST: return omni_launch_resize_nearest(op, s);
    case OMNI_OP_LETTERBOX: return omni_launch_letterbox(op, s);
    case OMNI_OP_DETECT_DECODE: return omni_launch_detect_decode(op, s);
    case OMNI_OP_NMS: return omni_launch_nms(op, s);
    case OMNI_OP_DWCONV3: return omni_launch_dwconv3(op, s);
    case OMNI_OP_LAYERNORM: return omni_launch_layernorm(op, s);
    case OMNI_OP_DWCONV3_LN: return omni_launch_dwconv3_ln(op, s);
    case OMNI_OP_ATTN_ROWS: case OMNI_OP_CHAN_ATTN: case OMNI_OP_ATTN_DECODE: return omni_launch_attention(op, s);
    case OMNI_OP_PROJ_PREP: case OMNI_OP_ASSEMBLE: case OMNI_OP_EMBED_STEP: case OMNI_OP_GREEDY_STEP:
    case OMNI_OP_CROP_RESIZE: return omni_launch_misc(op, s);
    default:
      omni_set_error("unknown op kind %d", op->kind);
      return OMNI_E_ARG;
  }
}

// ---------------------------------------------------------------------------------------------
// OMNI_CHECK_PTRS: turn a bad address into OMNI_E_ARG instead of a GPU memory-access fault (which aborts the process and, in a
// test run, hides every later test).  Every non-null p[k] of an op must lie in a device (or managed / host-registered) allocation
// known to the HIP runtime, and the byte range the op will touch from it must end inside that allocation
// (hipMemGetAddressRange; with a caching allocator the allocation is the allocator's segment — a wild pointer or a range that
// runs off the segment is caught, a neighbour inside the same segment is not).  Ranges are computed for the op kinds that hold
// almost all of a plan's launches (conv / GEMM family, pools / resizes, LayerNorm, depthwise conv, split-convert, fused FFN,
// detect-decode, NMS); for the other kinds (attention, decode step, crop, hand-off, PNG) the first byte is checked.  On by default wherever a device is present — omni_op_launch (the single-op path is
// never hot) and omni_plan_create (once per plan) — OMNI_CHECK_PTRS=0 turns it off; without a device there is nothing to check against.
static bool check_ptrs_enabled() {
  const char* e = getenv("OMNI_CHECK_PTRS");
  if (e && e[0] == '0') return false;
  // initialised once, thread-safely (omni_op_launch is entered from web-server workers and pipeline helper threads)
  static const int ndev = [] { int n = 0; const bool ok = hipGetDeviceCount(&n) == hipSuccess; if (!ok) (void)hipGetLastError(); return ok ? n : 0; }();
  return ndev > 0;
}

static void op_extents(const omni_op_t* op, long long ext[8]) {
  for (int k = 0; k < 8; ++k) ext[k] = 1;
  const long long esz = op->dtype == OMNI_F32 ? 4 : 2;
  const int* i = op->i;
  auto span = [&](long long rows, long long ld, long long coff, long long c) { return rows > 0 ? ((rows - 1) * ld + coff + c) * esz : 1; };
  switch (op->kind) {
    case OMNI_OP_CONV: {
      const long long B = i[0], H = i[1], W = i[2], Cin = i[3], Cout = i[12], M = B * i[10] * i[11], K = (long long)i[6] * i[7] * Cin;
      ext[0] = span(B * H * W, i[4], i[5], i[25] ? i[4] : Cin);    // row-patch mode: "Cin" = 8 pixels of ldi channels, never read beyond a row
      ext[1] = Cout * K * (i[20] ? 4 : esz);                       // split formats: two f16 halves per weight
      ext[2] = Cout * 4;
      ext[3] = span(M, i[16], i[17], Cout);
      ext[4] = span(M, i[13], i[14], Cout);
      ext[5] = i[19] > 0 ? (long long)i[19] * 1024 : 1;
      if (i[24] > 0) ext[6] = (long long)i[24] * 4;          // split-K arrival counters (in-launch combine); p6 is ignored when i24 == 0
      break;
    }
    case OMNI_OP_MLP_FUSED: {
      const long long rows = (long long)i[0] * i[1], C = i[3], hid = i[12];
      ext[0] = span(rows, i[4], i[5], C); ext[1] = hid * C * 4; ext[2] = hid * 4; ext[3] = span(rows, i[16], i[17], C);
      ext[4] = span(rows, i[13], i[14], C); ext[5] = hid * C * 4; ext[6] = C * 4;
      break;
    }
    case OMNI_OP_SPLIT_CONVERT: {
      const long long rows = (long long)i[0] * i[1];
      ext[0] = span(rows, i[4], i[5], i[3]); ext[4] = span(rows, i[13], i[14], i[3]);
      break;
    }
    case OMNI_OP_AVGPOOL2: case OMNI_OP_MAXPOOL: case OMNI_OP_RESIZE_NEAREST: {
      const long long B = i[0], Ho = op->kind == OMNI_OP_AVGPOOL2 ? i[1] - 1 : i[10], Wo = op->kind == OMNI_OP_AVGPOOL2 ? i[2] - 1 : i[11];
      ext[0] = span(B * i[1] * i[2], i[4], i[5], i[3]); ext[4] = span(B * Ho * Wo, i[13], i[14], i[3]);
      if (op->kind == OMNI_OP_RESIZE_NEAREST && i[17] > 1) {                      // CBFuse: sources 1..4 = p1, p2, p3, p5
        static const int pidx[4] = {1, 2, 3, 5};
        static const int base[4][4] = {{19, 20, 21, 22}, {23, 24, 25, 26}, {27, 28, 29, 30}, {7, 12, 15, 16}};
        for (int k = 0; k + 1 < i[17] && k < 4; ++k)
          ext[pidx[k]] = span(B * i[base[k][0]] * i[base[k][1]], i[base[k][2]], i[base[k][3]], i[3]);
      }
      break;
    }
    case OMNI_OP_NMS: {
      // p0 cand[frames][cap] p1 count[frames] p2 sorted[frames][cap + 1] p3 mask u64[cap * ceil(cap / 64)] p4..p7 outputs per frame
      const long long cap = i[0], md = i[1], fr = i[4] > 0 ? i[4] : 1;
      ext[0] = fr * cap * 32; ext[1] = fr * 4; ext[2] = fr * (cap + 1) * 32; ext[3] = cap * ((cap + 63) / 64) * 8;
      ext[4] = fr * md * 16; ext[5] = fr * md * 4; ext[6] = fr * md * 4; ext[7] = fr * 4;
      break;
    }
    case OMNI_OP_DETECT_DECODE: {
      // p0..p2 class heads, p3..p5 box heads of strides 8 / 16 / 32 over a TH x TW input; p6 cand[cap] p7 count
      for (int l = 0; l < 3; ++l) {
        const long long hw = (long long)(i[1] / (8 << l)) * (i[2] / (8 << l));
        ext[l] = span(hw, i[3 + l], i[13 + l], i[0]);
        ext[3 + l] = span(hw, i[6 + l], i[16 + l], i[12] ? 4 : 64);
      }
      ext[6] = (long long)i[9] * 32; ext[7] = 4;
      break;
    }
    case OMNI_OP_LAYERNORM: {
      const long long rows = (long long)i[0] * i[1], C = i[3];
      ext[0] = rows * C * esz; ext[1] = (long long)(i[5] > 0 ? i[5] : 1) * C * esz; ext[2] = C * 4; ext[3] = C * 4;
      ext[4] = rows * C * esz; ext[5] = rows * C * esz;
      break;
    }
    case OMNI_OP_DWCONV3: case OMNI_OP_DWCONV3_LN: {
      const long long n = (long long)i[0] * i[1] * i[2] * i[3] * esz, C = i[3];
      ext[0] = n; ext[1] = 9 * C * esz; ext[2] = C * 4; ext[4] = n;
      if (op->kind == OMNI_OP_DWCONV3_LN) { ext[3] = n; ext[5] = C * 4; ext[6] = C * 4; }
      break;
    }
    default: break;
  }
  for (int k = 0; k < 8; ++k) if (ext[k] < 1) ext[k] = 1;
}

static int check_op_ptrs(const omni_op_t* op, int index) {
  long long ext[8];
  op_extents(op, ext);
  for (int k = 0; k < 8; ++k) {
    const void* ptr = op->p[k];
    if (!ptr) continue;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr);
    if (e != hipSuccess || !base) {
      (void)hipGetLastError();
      omni_set_error("op %d (kind %d): p[%d] = %p is not inside any device allocation (%s)", index, op->kind, k, ptr,
                     e != hipSuccess ? hipGetErrorString(e) : "no base");
      return OMNI_E_ARG;
    }
    const long long off = (const char*)ptr - (const char*)base;
    if (off < 0 || off + ext[k] > (long long)size) {
      omni_set_error("op %d (kind %d): p[%d] = %p + %lld bytes runs past its allocation [%p, +%zu)", index, op->kind, k, ptr, ext[k],
                     (void*)base, size);
      return OMNI_E_ARG;
    }
  }
  return OMNI_OK;
}

extern "C" int omni_op_launch(const omni_op_t* op, void* stream) {
  if (!op) { omni_set_error("omni_op_launch: null op"); return OMNI_E_ARG; }
  if (check_ptrs_enabled()) { int rc = check_op_ptrs(op, 0); if (rc) return rc; }
  return dispatch(op, (hipStream_t)stream);
}

struct omni_plan {
  std::vector<omni_op_t> ops;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

extern "C" int omni_plan_create(const omni_op_t* ops, int n_ops, omni_plan_t** out) {
  if (!ops || n_ops <= 0 || !out) { omni_set_error("omni_plan_create: bad arguments"); return OMNI_E_ARG; }
  for (int i = 0; i < n_ops; ++i) {
    if (ops[i].kind <= 0 || ops[i].kind >= OMNI_OP__COUNT) {
      omni_set_error("omni_plan_create: op %d has unknown kind %d", i, ops[i].kind);
      return OMNI_E_ARG;
    }
  }
  if (check_ptrs_enabled())
    for (int i = 0; i < n_ops; ++i) { int rc = check_op_ptrs(&ops[i], i); if (rc) return rc; }
  omni_plan* p = new omni_plan();
  p->ops.assign(ops, ops + n_ops);
  *out = p;
  return OMNI_OK;
}

extern "C" int omni_plan_num_ops(const omni_plan_t* plan) { return plan ? (int)plan->ops.size() : 0; }

extern "C" int omni_plan_run(omni_plan_t* plan, void* stream) {
  if (!plan) { omni_set_error("omni_plan_run: null plan"); return OMNI_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  static const bool trace = getenv("OMNI_PLAN_TRACE") != nullptr;      // debugging aid: synchronise and name every op (eager replays only)
  for (size_t i = 0; i < plan->ops.size(); ++i) {
    if (trace) { fprintf(stderr, "[omni] op %zu/%zu kind %d ...", i, plan->ops.size(), plan->ops[i].kind); fflush(stderr); }
    int rc = dispatch(&plan->ops[i], s);
    if (trace) { hipError_t e = hipStreamSynchronize(s); fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); fflush(stderr); }
    if (rc) {
      char prev[400];
      strncpy(prev, g_err, sizeof(prev) - 1); prev[sizeof(prev) - 1] = 0;
      omni_set_error("plan op %zu (kind %d): %s", i, plan->ops[i].kind, prev);
      return rc;
    }
  }
  return OMNI_OK;
}

extern "C" int omni_plan_capture(omni_plan_t* plan, void* stream) {
  if (!plan) { omni_set_error("omni_plan_capture: null plan"); return OMNI_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  if (!s) { omni_set_error("omni_plan_capture: needs a non-default stream"); return OMNI_E_ARG; }
  if (plan->exec) { hipGraphExecDestroy(plan->exec); plan->exec = nullptr; }
  if (plan->graph) { hipGraphDestroy(plan->graph); plan->graph = nullptr; }
  OMNI_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  int rc = omni_plan_run(plan, s);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(s, &g);
  if (rc) { if (g) hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) { omni_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return OMNI_E_HIP; }
  plan->graph = g;
  OMNI_HIP_CHECK(hipGraphInstantiate(&plan->exec, plan->graph, nullptr, nullptr, 0));
  return OMNI_OK;
}

extern "C" int omni_plan_replay(omni_plan_t* plan, void* stream) {
  if (!plan) { omni_set_error("omni_plan_replay: null plan"); return OMNI_E_ARG; }
  if (!plan->exec) return omni_plan_run(plan, stream);
  OMNI_HIP_CHECK(hipGraphLaunch(plan->exec, (hipStream_t)stream));
  return OMNI_OK;
}

extern "C" void omni_plan_destroy(omni_plan_t* plan) {
  if (!plan) return;
  if (plan->exec) hipGraphExecDestroy(plan->exec);
  if (plan->graph) hipGraphDestroy(plan->graph);
  delete plan;
}

extern "C" int omni_plan_time(omni_plan_t* plan, void* stream, int iters, float* ms) {
  if (!plan || iters <= 0 || !ms) { omni_set_error("omni_plan_time: bad arguments"); return OMNI_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  OMNI_HIP_CHECK(hipEventCreate(&e0));
  OMNI_HIP_CHECK(hipEventCreate(&e1));
  OMNI_HIP_CHECK(hipEventRecord(e0, s));
  int rc = OMNI_OK;
  for (int i = 0; i < iters && rc == OMNI_OK; ++i) rc = omni_plan_replay(plan, stream);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float t = 0.f;
  hipEventElapsedTime(&t, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (rc) return rc;
  *ms = t / iters;
  return OMNI_OK;
}

// Per-op device time of one eager replay: an event in front of every op and one behind the last, all on `stream`, so each
// op is timed in its real sequence (caches as its predecessors left them, neighbours interleaved) — what a kernel trace
// of the same replay reports per launch.  h_ms[i] = milliseconds between the events around op i (ops that launch several
// kernels, e.g. a split-K conv + its reduce, are timed as one).
extern "C" int omni_plan_profile(omni_plan_t* plan, void* stream, float* h_ms) {
  if (!plan || !h_ms) { omni_set_error("omni_plan_profile: bad arguments"); return OMNI_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const size_t n = plan->ops.size();
  std::vector<hipEvent_t> ev(n + 1, nullptr);
  int rc = OMNI_OK;
  for (size_t i = 0; i <= n && rc == OMNI_OK; ++i)
    if (hipEventCreate(&ev[i]) != hipSuccess) { omni_set_error("omni_plan_profile: hipEventCreate failed"); rc = OMNI_E_HIP; }
  for (size_t i = 0; i < n && rc == OMNI_OK; ++i) {
    hipEventRecord(ev[i], s);
    rc = dispatch(&plan->ops[i], s);
  }
  if (rc == OMNI_OK) {
    hipEventRecord(ev[n], s);
    if (hipEventSynchronize(ev[n]) != hipSuccess) { omni_set_error("omni_plan_profile: synchronize failed"); rc = OMNI_E_HIP; }
  }
  for (size_t i = 0; i < n && rc == OMNI_OK; ++i) {
    float t = 0.f;
    hipEventElapsedTime(&t, ev[i], ev[i + 1]);
    h_ms[i] = t;
  }
  for (hipEvent_t e : ev) if (e) hipEventDestroy(e);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// Pillow precompute_coeffs + normalize_coeffs_8bpc (host, double precision).
static double sinc_filter(double x) {
  if (x == 0.0) return 1.0;
  x = x * M_PI;
  return sin(x) / x;
}
static double lanczos_filter(double x) {
  if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
  return 0.0;
}
static double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

extern "C" int omni_resample_coeffs(int in_size, int out_size, int filter, int32_t* h_bounds, int32_t* h_coef) {
  if (in_size <= 0 || out_size <= 0 || (filter != 0 && filter != 1)) {
    omni_set_error("omni_resample_coeffs: bad arguments");
    return OMNI_E_ARG;
  }
  double (*f)(double) = filter == 0 ? lanczos_filter : bicubic_filter;
  double fsupport = filter == 0 ? 3.0 : 2.0;
  double scale = (double)in_size / out_size;
  double filterscale = scale < 1.0 ? 1.0 : scale;
  double support = fsupport * filterscale;
  int ksize = (int)ceil(support) * 2 + 1;
  if (!h_bounds || !h_coef) return ksize;
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    double center = (xx + 0.5) * scale;
    double ww = 0.0;
    double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double w = f((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = 0; x < ksize; ++x) {
      double v = x < xmax ? k[x] : 0.0;
      h_coef[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22));
    }
    h_bounds[xx * 2 + 0] = xmin;
    h_bounds[xx * 2 + 1] = xmax;
  }
  return ksize;
}
