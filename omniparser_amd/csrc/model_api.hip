// Model-level entry points of libomni_amd.so (include/omni_amd.h: omni_model_*, omni_detector_*, omni_captioner_*): the two models
// of the hot path behind plain C calls, for hosts without Python / torch.  A model is a PLAN BUNDLE written by
// omniparser_amd/bundle.py — the op lists the Python graph builders produce, their device buffers, constants and named I/O
// tensors — loaded here: allocate, upload, relocate, capture as hipGraphs on a stream owned by the model.
//   omni_detector_infer    stands behind ref:util/yolov9.py:115-136 (YOLOv9Detector.predict on one batch)
//   omni_captioner_caption stands behind ref:util/utils.py:88-132 (crop, cv2.resize, processor, model.generate, greedy ids)
// Host code only; every device operation goes through the same kernels / plan executor as the Python host's calls.
#include "omni_internal.h"

#include <stdlib.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>

struct omni_model {
  std::vector<void*> tensors;
  std::vector<size_t> sizes;
  std::map<std::string, omni_plan_t*> plans;
  struct Named { int tensor; long long off, nbytes; };
  std::map<std::string, Named> named;
  std::map<std::string, long long> ints;
  hipStream_t stream = nullptr;
  bool graphs = false;
};

namespace {

struct Reader {
  FILE* f;
  bool ok = true;
  template <typename T> T get() {
    T v{};
    if (fread(&v, sizeof(T), 1, f) != 1) ok = false;
    return v;
  }
  std::string name() {
    char b[33] = {0};
    if (fread(b, 1, 32, f) != 32) ok = false;
    return std::string(b);
  }
};

void* named_ptr(const omni_model* m, const char* name, long long* nbytes = nullptr) {
  auto it = m->named.find(name);
  if (it == m->named.end()) return nullptr;
  if (nbytes) *nbytes = it->second.nbytes;
  return (unsigned char*)m->tensors[it->second.tensor] + it->second.off;
}

long long model_int(const omni_model* m, const char* name, long long dflt = 0) {
  auto it = m->ints.find(name);
  return it == m->ints.end() ? dflt : it->second;
}

float bits_f32(long long v) {
  int32_t b = (int32_t)v;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

// every named tensor / integer an entry point uses must exist (and be large enough) BEFORE the first launch: a stale, truncated or
// mismatched bundle is OMNI_E_ARG at create time, never an out-of-bounds device write or a null-pointer HIP call later
struct NeedT { const char* name; long long min_bytes; };
int require_named(const omni_model* m, const char* who, const NeedT* need, size_t n_need, const char* const* ints, size_t n_ints) {
  for (size_t k = 0; k < n_ints; ++k)
    if (!m->ints.count(ints[k])) { omni_set_error("%s: the bundle has no integer '%s'", who, ints[k]); return OMNI_E_ARG; }
  for (size_t k = 0; k < n_need; ++k) {
    long long nb = 0;
    if (!named_ptr(m, need[k].name, &nb)) { omni_set_error("%s: the bundle has no tensor '%s'", who, need[k].name); return OMNI_E_ARG; }
    if (nb < need[k].min_bytes) {
      omni_set_error("%s: tensor '%s' holds %lld bytes, the bundle's own sizes need %lld", who, need[k].name, nb, need[k].min_bytes);
      return OMNI_E_ARG;
    }
  }
  return OMNI_OK;
}

int run_plan(omni_model* m, const char* name) {
  auto it = m->plans.find(name);
  if (it == m->plans.end()) { omni_set_error("model has no plan '%s'", name); return OMNI_E_ARG; }
  return m->graphs ? omni_plan_replay(it->second, m->stream) : omni_plan_run(it->second, m->stream);
}

}  // namespace

extern "C" void omni_model_destroy(omni_model_t* m) {
  if (!m) return;
  for (auto& kv : m->plans) omni_plan_destroy(kv.second);
  for (void* p : m->tensors) if (p) hipFree(p);
  if (m->stream) hipStreamDestroy(m->stream);
  delete m;
}

extern "C" int omni_model_load(const char* path, omni_model_t** out) {
  if (!path || !out) { omni_set_error("omni_model_load: bad arguments"); return OMNI_E_ARG; }
  FILE* f = fopen(path, "rb");
  if (!f) { omni_set_error("omni_model_load: cannot open %s", path); return OMNI_E_ARG; }
  Reader r{f};
  char magic[8];
  static_assert(OMNI_ABI_VERSION >= 0 && OMNI_ABI_VERSION <= 9, "one digit of the magic holds the ABI version");
  const char want[9] = {'O', 'M', 'N', 'I', 'P', 'L', 'N', (char)('0' + OMNI_ABI_VERSION), 0};
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, want, 7) != 0) { fclose(f); omni_set_error("omni_model_load: %s is not a plan bundle", path); return OMNI_E_ARG; }
  if (magic[7] != want[7]) {   // op slot meanings change with the ABI (ABI 2 redefined OMNI_OP_CONV i22): a bundle is only valid under the ABI that wrote it
    fclose(f);
    omni_set_error("omni_model_load: %s was exported under ABI %c, this library is ABI %d (re-export the bundle)", path, magic[7], OMNI_ABI_VERSION);
    return OMNI_E_ARG;
  }
  const uint32_t nt = r.get<uint32_t>(), np = r.get<uint32_t>(), nn = r.get<uint32_t>(), ni = r.get<uint32_t>();
  if (!r.ok || nt > (1u << 20) || np > 1024u || nn > 4096u || ni > 4096u) {
    fclose(f);
    omni_set_error("omni_model_load: %s has an implausible header (%u tensors, %u plans, %u names, %u ints)", path, nt, np, nn, ni);
    return OMNI_E_ARG;
  }
  omni_model* m = new omni_model();
  struct TRec { uint64_t nbytes; uint32_t role, pad; uint64_t off; };
  std::vector<TRec> recs(nt);
  for (auto& t : recs) { t.nbytes = r.get<uint64_t>(); t.role = r.get<uint32_t>(); t.pad = r.get<uint32_t>(); t.off = r.get<uint64_t>(); }
  struct PRec { int32_t t; long long off; };
  struct ORec { int32_t kind, dtype; PRec p[8]; int32_t i[32]; float fl[8]; };
  std::vector<std::pair<std::string, std::vector<ORec>>> plans(np);
  for (auto& pl : plans) {
    pl.first = r.name();
    const uint32_t n = r.get<uint32_t>();
    if (!r.ok || n > (1u << 20)) { r.ok = false; break; }
    pl.second.resize(n);
    for (auto& o : pl.second) {
      o.kind = r.get<int32_t>(); o.dtype = r.get<int32_t>();
      for (auto& p : o.p) { p.t = r.get<int32_t>(); p.off = r.get<long long>(); }
      for (auto& v : o.i) v = r.get<int32_t>();
      for (auto& v : o.fl) v = r.get<float>();
    }
  }
  for (uint32_t k = 0; k < nn && r.ok; ++k) {
    std::string nm = r.name();
    omni_model::Named nd; nd.tensor = r.get<int32_t>(); nd.off = r.get<long long>(); nd.nbytes = r.get<long long>();
    m->named[nm] = nd;
  }
  for (uint32_t k = 0; k < ni && r.ok; ++k) { std::string nm = r.name(); m->ints[nm] = r.get<long long>(); }
  int rc = OMNI_OK;
  auto fail = [&](int code) { fclose(f); omni_model_destroy(m); return code; };
  if (!r.ok) { omni_set_error("omni_model_load: %s is truncated", path); return fail(OMNI_E_ARG); }
  for (const auto& kv : m->named) {                         // a named tensor is a byte range INSIDE one tensor of the bundle
    const omni_model::Named& nd = kv.second;
    if (nd.tensor < 0 || (uint32_t)nd.tensor >= nt || nd.off < 0 || nd.nbytes < 0 || (uint64_t)nd.off + (uint64_t)nd.nbytes > recs[nd.tensor].nbytes) {
      omni_set_error("omni_model_load: named tensor '%s' (tensor %d, offset %lld, %lld bytes) is out of range", kv.first.c_str(), nd.tensor,
                     nd.off, nd.nbytes);
      return fail(OMNI_E_ARG);
    }
  }
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { omni_set_error("omni_model_load: hipStreamCreate failed"); return fail(OMNI_E_HIP); }
  // tensors: allocate, zero, upload constants (staged through a 16 MiB host buffer)
  m->tensors.assign(nt, nullptr);
  m->sizes.resize(nt);
  std::vector<unsigned char> stage(16u << 20);
  for (uint32_t k = 0; k < nt; ++k) {
    m->sizes[k] = recs[k].nbytes;
    if (hipMalloc(&m->tensors[k], recs[k].nbytes ? recs[k].nbytes : 4) != hipSuccess) {
      omni_set_error("omni_model_load: hipMalloc of %llu bytes failed (tensor %u)", (unsigned long long)recs[k].nbytes, k);
      return fail(OMNI_E_HIP);
    }
    if (recs[k].role != 2) {
      if (hipMemset(m->tensors[k], 0, recs[k].nbytes) != hipSuccess) { omni_set_error("omni_model_load: hipMemset failed"); return fail(OMNI_E_HIP); }
      continue;
    }
    if (fseek(f, (long)recs[k].off, SEEK_SET) != 0) { omni_set_error("omni_model_load: bad data offset"); return fail(OMNI_E_ARG); }
    for (uint64_t done = 0; done < recs[k].nbytes;) {
      const size_t n = (size_t)std::min<uint64_t>(stage.size(), recs[k].nbytes - done);
      if (fread(stage.data(), 1, n, f) != n) { omni_set_error("omni_model_load: %s is truncated (constants)", path); return fail(OMNI_E_ARG); }
      if (hipMemcpy((unsigned char*)m->tensors[k] + done, stage.data(), n, hipMemcpyHostToDevice) != hipSuccess) {
        omni_set_error("omni_model_load: upload failed"); return fail(OMNI_E_HIP);
      }
      done += n;
    }
  }
  fclose(f);
  f = nullptr;
  auto fail2 = [&](int code) { omni_model_destroy(m); return code; };
  // plans: relocate the pointers, build, warm up, capture
  for (auto& pl : plans) {
    std::vector<omni_op_t> ops(pl.second.size());
    for (size_t j = 0; j < ops.size(); ++j) {
      const ORec& o = pl.second[j];
      memset(&ops[j], 0, sizeof(omni_op_t));
      ops[j].kind = o.kind; ops[j].dtype = o.dtype;
      for (int q = 0; q < 8; ++q) {
        if (o.p[q].t < 0) continue;
        if ((uint32_t)o.p[q].t >= nt || o.p[q].off < 0 || (uint64_t)o.p[q].off >= std::max<uint64_t>(recs[o.p[q].t].nbytes, 1)) {
          omni_set_error("omni_model_load: plan '%s' op %zu pointer %d is out of range", pl.first.c_str(), j, q);
          return fail2(OMNI_E_ARG);
        }
        ops[j].p[q] = (unsigned char*)m->tensors[o.p[q].t] + o.p[q].off;
      }
      memcpy(ops[j].i, o.i, sizeof(o.i));
      memcpy(ops[j].f, o.fl, sizeof(o.fl));
    }
    omni_plan_t* plan = nullptr;
    rc = omni_plan_create(ops.data(), (int)ops.size(), &plan);
    if (rc) return fail2(rc);
    m->plans[pl.first] = plan;
  }
  const char* g = getenv("OMNI_HIPGRAPH");
  m->graphs = !(g && atoi(g) == 0);
  if (m->graphs) {
    for (auto& kv : m->plans) {
      rc = omni_plan_run(kv.second, m->stream);                     // warm-up (module load) outside capture
      if (rc) return fail2(rc);
      if (hipStreamSynchronize(m->stream) != hipSuccess) { omni_set_error("omni_model_load: warm-up of plan '%s' failed", kv.first.c_str()); return fail2(OMNI_E_HIP); }
      rc = omni_plan_capture(kv.second, m->stream);
      if (rc) return fail2(rc);
    }
    hipStreamSynchronize(m->stream);
  }
  *out = m;
  return OMNI_OK;
}

extern "C" int omni_model_int(const omni_model_t* m, const char* name, long long* value) {
  if (!m || !name || !value) { omni_set_error("omni_model_int: bad arguments"); return OMNI_E_ARG; }
  auto it = m->ints.find(name);
  if (it == m->ints.end()) { omni_set_error("model has no integer '%s'", name); return OMNI_E_ARG; }
  *value = it->second;
  return OMNI_OK;
}

extern "C" int omni_model_tensor(const omni_model_t* m, const char* name, void** d_ptr, long long* nbytes) {
  if (!m || !name || !d_ptr) { omni_set_error("omni_model_tensor: bad arguments"); return OMNI_E_ARG; }
  void* p = named_ptr(m, name, nbytes);
  if (!p) { omni_set_error("model has no tensor '%s'", name); return OMNI_E_ARG; }
  *d_ptr = p;
  return OMNI_OK;
}

extern "C" int omni_model_run(omni_model_t* m, const char* plan_name) {
  if (!m || !plan_name) { omni_set_error("omni_model_run: bad arguments"); return OMNI_E_ARG; }
  int rc = run_plan(m, plan_name);
  if (rc) return rc;
  OMNI_HIP_CHECK(hipStreamSynchronize(m->stream));
  return OMNI_OK;
}

// ---------------------------------------------------------------------------------------------------- detector
extern "C" int omni_detector_create(const char* bundle_path, omni_model_t** out) {
  int rc = omni_model_load(bundle_path, out);
  if (rc) return rc;
  if (model_int(*out, "model") != 1 || !(*out)->plans.count("detect")) {
    omni_model_destroy(*out); *out = nullptr;
    omni_set_error("omni_detector_create: %s is not a detector bundle", bundle_path);
    return OMNI_E_ARG;
  }
  static const char* const ints[] = {"batch", "img_w", "img_h", "max_det"};
  rc = require_named(*out, "omni_detector_create", nullptr, 0, ints, 4);
  if (rc == OMNI_OK) {
    const long long batch = model_int(*out, "batch"), W = model_int(*out, "img_w"), H = model_int(*out, "img_h"), md = model_int(*out, "max_det");
    if (batch <= 0 || W <= 0 || H <= 0 || md <= 0 || batch > 4096 || W > 65536 || H > 65536 || md > 65536) {
      omni_set_error("omni_detector_create: implausible sizes (batch %lld, %lld x %lld, max_det %lld)", batch, W, H, md);
      rc = OMNI_E_ARG;
    } else {
      const NeedT need[] = {{"img", batch * H * W * 3}, {"out_count", batch * 4}, {"out_boxes", batch * md * 16}, {"out_scores", batch * md * 4},
                            {"out_cls", batch * md * 4}};
      rc = require_named(*out, "omni_detector_create", need, 5, nullptr, 0);
    }
  }
  if (rc) { omni_model_destroy(*out); *out = nullptr; return rc; }
  return OMNI_OK;
}

extern "C" int omni_detector_infer(omni_model_t* det, const uint8_t* images_rgb, int n_images, int on_device, float* h_boxes, float* h_scores,
                                   int32_t* h_classes, int32_t* h_counts) {
  if (!det || !images_rgb || !h_boxes || !h_counts || n_images <= 0) { omni_set_error("omni_detector_infer: bad arguments"); return OMNI_E_ARG; }
  const long long batch = model_int(det, "batch"), W = model_int(det, "img_w"), H = model_int(det, "img_h"), md = model_int(det, "max_det");
  if (n_images > batch) { omni_set_error("omni_detector_infer: %d images, the bundle was exported for a batch of %lld", n_images, batch); return OMNI_E_ARG; }
  void* img = named_ptr(det, "img");
  OMNI_HIP_CHECK(hipMemcpyAsync(img, images_rgb, (size_t)n_images * H * W * 3, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, det->stream));
  int rc = run_plan(det, "detect");
  if (rc) return rc;
  OMNI_HIP_CHECK(hipMemcpyAsync(h_counts, named_ptr(det, "out_count"), (size_t)n_images * 4, hipMemcpyDeviceToHost, det->stream));
  OMNI_HIP_CHECK(hipMemcpyAsync(h_boxes, named_ptr(det, "out_boxes"), (size_t)n_images * md * 16, hipMemcpyDeviceToHost, det->stream));
  if (h_scores) OMNI_HIP_CHECK(hipMemcpyAsync(h_scores, named_ptr(det, "out_scores"), (size_t)n_images * md * 4, hipMemcpyDeviceToHost, det->stream));
  if (h_classes) OMNI_HIP_CHECK(hipMemcpyAsync(h_classes, named_ptr(det, "out_cls"), (size_t)n_images * md * 4, hipMemcpyDeviceToHost, det->stream));
  OMNI_HIP_CHECK(hipStreamSynchronize(det->stream));
  return OMNI_OK;
}

// ---------------------------------------------------------------------------------------------------- captioner
extern "C" int omni_captioner_create(const char* bundle_path, omni_model_t** out) {
  int rc = omni_model_load(bundle_path, out);
  if (rc) return rc;
  if (model_int(*out, "model") != 2 || !(*out)->plans.count("encode") || !(*out)->plans.count("step")) {
    omni_model_destroy(*out); *out = nullptr;
    omni_set_error("omni_captioner_create: %s is not a captioner bundle", bundle_path);
    return OMNI_E_ARG;
  }
  static const char* const ints[] = {"capacity", "R", "T", "max_new", "start_token", "dtype", "ksize", "ldo",
                                     "mean0", "mean1", "mean2", "std0", "std1", "std2"};
  rc = require_named(*out, "omni_captioner_create", nullptr, 0, ints, sizeof(ints) / sizeof(ints[0]));
  if (rc == OMNI_OK) {
    const long long B = model_int(*out, "capacity"), R = model_int(*out, "R"), T = model_int(*out, "T"), ks = model_int(*out, "ksize"),
                    ldo = model_int(*out, "ldo"), dt = model_int(*out, "dtype"), esz = dt == OMNI_F32 ? 4 : 2;
    if (B <= 0 || R <= 0 || T <= 0 || ks < 0 || (ks == 0) != (R == 64) || ldo < 3 || B > 65536 || R > 8192 || T > 4096 || ks > 64 || (dt != OMNI_F32 && dt != OMNI_F16) ||
        model_int(*out, "max_new") < 0 || model_int(*out, "max_new") >= T) {
      omni_set_error("omni_captioner_create: implausible sizes (capacity %lld, R %lld, T %lld, ksize %lld, ldo %lld, dtype %lld)", B, R, T, ks, ldo, dt);
      rc = OMNI_E_ARG;
    } else {
      // R == 64: the crop op's bicubic pass is skipped (ksize 0) and its tables / row buffer are absent
      const NeedT need[] = {{"ids", B * T * 4}, {"finished", B * 4}, {"step", 4}, {"boxes", B * 16}, {"c64", B * 64 * 64 * 3},
                            {"x_in", B * R * R * ldo * esz}, {"lut", 256 * 4},
                            {"tmp", B * 64 * R * 3}, {"bic_bounds", R * 2 * 4}, {"bic_coef", R * ks * 4}};
      rc = require_named(*out, "omni_captioner_create", need, ks ? 10 : 7, nullptr, 0);
    }
  }
  if (rc) { omni_model_destroy(*out); *out = nullptr; return rc; }
  return OMNI_OK;
}

extern "C" int omni_captioner_caption(omni_model_t* cap, const uint8_t* image_rgb, int on_device, int img_h, int img_w, const int32_t* h_boxes_px,
                                      int n, int32_t* h_ids) {
  if (!cap || !image_rgb || !h_boxes_px || !h_ids || n < 0 || img_h <= 0 || img_w <= 0) { omni_set_error("omni_captioner_caption: bad arguments"); return OMNI_E_ARG; }
  const int B = (int)model_int(cap, "capacity"), R = (int)model_int(cap, "R"), T = (int)model_int(cap, "T"), max_new = (int)model_int(cap, "max_new");
  const int start = (int)model_int(cap, "start_token");
  const uint8_t* d_img = image_rgb;
  void* owned = nullptr;
  if (!on_device) {
    OMNI_HIP_CHECK(hipMalloc(&owned, (size_t)img_h * img_w * 3));
    if (hipMemcpyAsync(owned, image_rgb, (size_t)img_h * img_w * 3, hipMemcpyHostToDevice, cap->stream) != hipSuccess) { hipFree(owned); omni_set_error("omni_captioner_caption: image upload failed"); return OMNI_E_HIP; }
    d_img = (const uint8_t*)owned;
  }
  std::vector<int32_t> ids0((size_t)B * T, 0);
  for (int b = 0; b < B; ++b) ids0[(size_t)b * T] = start;
  int rc = OMNI_OK;
  for (int s = 0; s < n && rc == OMNI_OK; s += B) {
    const int m = n - s < B ? n - s : B;
    // state reset (florence.py::_CaptionPlans.reset), crop rectangles, crop pre-processing, encode, max_new decode steps
    if (hipMemcpyAsync(named_ptr(cap, "ids"), ids0.data(), ids0.size() * 4, hipMemcpyHostToDevice, cap->stream) != hipSuccess ||
        hipMemsetAsync(named_ptr(cap, "finished"), 0, (size_t)B * 4, cap->stream) != hipSuccess ||
        hipMemsetAsync(named_ptr(cap, "step"), 0, 4, cap->stream) != hipSuccess ||
        hipMemcpyAsync(named_ptr(cap, "boxes"), h_boxes_px + (size_t)s * 4, (size_t)m * 16, hipMemcpyHostToDevice, cap->stream) != hipSuccess) {
      omni_set_error("omni_captioner_caption: state reset failed: %s", hipGetErrorString(hipGetLastError()));
      rc = OMNI_E_HIP;
      break;
    }
    omni_op_t op;
    memset(&op, 0, sizeof(op));
    op.kind = OMNI_OP_CROP_RESIZE;
    op.dtype = (int32_t)model_int(cap, "dtype");
    op.p[0] = (void*)d_img; op.p[1] = named_ptr(cap, "boxes"); op.p[2] = named_ptr(cap, "c64"); op.p[3] = named_ptr(cap, "tmp");
    op.p[4] = named_ptr(cap, "x_in"); op.p[5] = named_ptr(cap, "bic_bounds"); op.p[6] = named_ptr(cap, "bic_coef"); op.p[7] = named_ptr(cap, "lut");
    op.i[0] = m; op.i[1] = img_h; op.i[2] = img_w; op.i[3] = R; op.i[4] = (int32_t)model_int(cap, "ksize"); op.i[13] = (int32_t)model_int(cap, "ldo");
    for (int c = 0; c < 3; ++c) {
      char k[8];
      snprintf(k, sizeof(k), "mean%d", c); op.f[c] = bits_f32(model_int(cap, k));
      snprintf(k, sizeof(k), "std%d", c); op.f[3 + c] = bits_f32(model_int(cap, k));
    }
    rc = omni_op_launch(&op, cap->stream);
    if (rc == OMNI_OK) rc = run_plan(cap, "encode");
    for (int t = 0; t < max_new && rc == OMNI_OK; ++t) rc = run_plan(cap, "step");
    if (rc == OMNI_OK && hipMemcpyAsync(h_ids + (size_t)s * T, named_ptr(cap, "ids"), (size_t)m * T * 4, hipMemcpyDeviceToHost, cap->stream) != hipSuccess) {
      omni_set_error("omni_captioner_caption: read-back failed"); rc = OMNI_E_HIP;
    }
    if (rc == OMNI_OK && hipStreamSynchronize(cap->stream) != hipSuccess) { omni_set_error("omni_captioner_caption: device error"); rc = OMNI_E_HIP; }
  }
  if (owned) { hipStreamSynchronize(cap->stream); hipFree(owned); }
  return rc;
}
