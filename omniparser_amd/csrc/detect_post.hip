// Detector post-processing on device: DFL + anchor decode + sigmoid + max-class + threshold +
// un-letterbox + compaction, then torchvision-exact class-aware greedy NMS.
//
// Follows, operation for operation (f32, no FMA contraction, IEEE divide):
//   ref:util/yolov9.py:89-108   _decode  (dist*stride, anchors=(grid+0.5)*stride, xyxy, sigmoid)
//   ref:util/yolov9.py:123-129  max over classes (first index on ties), scores > conf (strict),
//                               (x - pad_left)/scale, (y - pad_top)/scale
//   ref:util/yolov9.py:131      torchvision.ops.batched_nms (CPU dispatch: coordinate-offset trick
//                               when boxes.numel() <= 4000, per-class otherwise) [:max_det]
//   ref:util/yolov9.py:134-135  clamp AFTER nms
// The DFL expectation (softmax over 16 bins . arange(16)) lives inside the reference's TorchScript
// blob (SURVEY App. B head); here it is fused into the decode so the 64-channel box logits are read
// from HBM exactly once.
//
// wave64 notes: compaction uses one atomic per passing anchor (few % of anchors pass).
// NMS, up to 2048 candidates per frame (every 640x640 input: ~1 400): ONE 1024-thread workgroup per frame (round 5,
// nms_block_kernel) — bitonic sort of 64-bit keys (score desc, anchor asc == torch's stable descending sort) in LDS, the
// candidates' offset boxes in LDS, every 64-box block's own 64x64 IoU mask built by the 16 waves in parallel, then per block: wave 0
// resolves the block greedily with v_readlane, and all 16 waves apply its kept boxes to every later candidate; it stops at max_det keeps.
// No N x N mask in HBM, no O(N^2) rank pass, no serial chain of dependent global loads (the round-1 path cost 547 us per frame).
// NMS, more candidates (1088x1920 inputs: ~9 000): the round-1 path — O(N^2) rank kernel, IoU predicate in 64x64 tiles into
// 64-bit masks (one u64 per lane = one row), single-wave greedy pass that resolves a 64-box block in registers with
// v_readlane and ORs the kept rows' masks into an LDS-resident suppression vector.  Both paths evaluate the SAME f32
// expressions in the same operand order, so their keep lists are bit-identical (tests: known answers + random clouds on both).
#include "omni_internal.h"

#pragma clang fp contract(off)

namespace {

struct Cand {
  float x1, y1, x2, y2;
  float score;
  int cls;
  int anchor;
  int pad;
};
static_assert(sizeof(Cand) == 32, "omni_cand_t layout");

struct DecodeArgs {
  const void* cls[3]; const void* box[3];
  int ldc[3], ldb[3], coffc[3], coffb[3];
  int hs[3], ws[3], astart[4];
  int nc, cap, pad_left, pad_top, dist_reduced;
  float conf, scale;
  Cand* cand; int* count;
};

template <typename T>
__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.astart[3]) return;
  int lvl = idx >= a.astart[2] ? 2 : (idx >= a.astart[1] ? 1 : 0);
  int local = idx - a.astart[lvl];
  int stride = 8 << lvl;
  int wsz = a.ws[lvl];
  int iy = local / wsz, ix = local - iy * wsz;
  const T* cl = reinterpret_cast<const T*>(a.cls[lvl]) + (long long)local * a.ldc[lvl] + a.coffc[lvl];
  // sigmoid then max, first index wins ties (torch.max on CPU)
  float best = -1.0f;
  int bestc = 0;
  for (int c = 0; c < a.nc; ++c) {
    float l = ElemTraits<T>::to_f32(cl[c]);
    float sgm = 1.0f / (1.0f + expf(-l));
    if (sgm > best) { best = sgm; bestc = c; }
  }
  if (!(best > a.conf)) return;
  const T* bx = reinterpret_cast<const T*>(a.box[lvl]) + (long long)local * a.ldb[lvl] + a.coffb[lvl];
  float d[4];
  if (a.dist_reduced) {
#pragma unroll
    for (int s = 0; s < 4; ++s) d[s] = ElemTraits<T>::to_f32(bx[s]);
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float v[16];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) { v[i] = ElemTraits<T>::to_f32(bx[s * 16 + i]); mx = fmaxf(mx, v[i]); }
      float sum = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { v[i] = expf(v[i] - mx); sum += v[i]; }
      float e = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) e += (v[i] / sum) * (float)i;
      d[s] = e;
    }
  }
  float fs = (float)stride;
  float ax = ((float)ix + 0.5f) * fs;
  float ay = ((float)iy + 0.5f) * fs;
  float x1 = ax - d[0] * fs, y1 = ay - d[1] * fs, x2 = ax + d[2] * fs, y2 = ay + d[3] * fs;
  float pl = (float)a.pad_left, pt = (float)a.pad_top;
  Cand c;
  c.x1 = (x1 - pl) / a.scale;
  c.y1 = (y1 - pt) / a.scale;
  c.x2 = (x2 - pl) / a.scale;
  c.y2 = (y2 - pt) / a.scale;
  c.score = best;
  c.cls = bestc;
  c.anchor = idx;
  c.pad = 0;
  int slot = atomicAdd(a.count, 1);
  if (slot < a.cap) a.cand[slot] = c;
}

// ---------------------------------------------------------------------------------------------
struct NmsArgs {
  const Cand* cand; const int* count; Cand* sorted; unsigned long long* mask;
  float* out_boxes; float* out_scores; int* out_cls; int* out_count;
  float* meta;   // [0] = max coordinate, [1] = use_trick flag (as float bits), lives after sorted[]
  int cap, max_det, img_w, img_h;
  float iou;
  int frames;      // nms_block_kernel: one workgroup per frame; frame f's buffers = base + f * (cap | cap + 1 | max_det | 1) records
};

constexpr int NMS_FAST_N = 2048;          // candidates the single-workgroup path holds in LDS
constexpr int NMS_FAST_ANCHOR_BITS = 21;  // key = ~score bits (32) | anchor (21) | slot (11)

__device__ __forceinline__ int clamp_count(const int* count, int cap) {
  int n = *count;
  return n < cap ? n : cap;
}

// rank sort: sorted[rank(i)] = cand[i], key = (score desc, anchor asc)
__global__ __launch_bounds__(256) void rank_kernel(NmsArgs a, int fast) {
  __shared__ float s_score[256];
  __shared__ int s_anchor[256];
  const int n = clamp_count(a.count, a.cap);
  if (fast && n <= NMS_FAST_N) return;          // nms_block_kernel took this frame
  for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
    int i = base + threadIdx.x;
    Cand me;
    float ms = 0.f; int ma = 0;
    if (i < n) { me = a.cand[i]; ms = me.score; ma = me.anchor; }
    int rank = 0;
    for (int t = 0; t < n; t += 256) {
      int j = t + threadIdx.x;
      __syncthreads();
      if (j < n) { s_score[threadIdx.x] = a.cand[j].score; s_anchor[threadIdx.x] = a.cand[j].anchor; }
      __syncthreads();
      int lim = n - t < 256 ? n - t : 256;
      for (int q = 0; q < lim; ++q) {
        float sj = s_score[q]; int aj = s_anchor[q];
        rank += (sj > ms || (sj == ms && aj < ma)) ? 1 : 0;
      }
    }
    if (i < n) a.sorted[rank] = me;
  }
}

// max coordinate + dispatch flag of torchvision.ops.batched_nms (CPU thresholds)
__global__ __launch_bounds__(256) void nms_prep_kernel(NmsArgs a, int fast) {
  __shared__ float s_max[256];
  const int n = clamp_count(a.count, a.cap);
  if (fast && n <= NMS_FAST_N) return;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) {
    Cand c = a.cand[i];
    m = fmaxf(m, fmaxf(fmaxf(c.x1, c.y1), fmaxf(c.x2, c.y2)));
  }
  s_max[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.meta[0] = s_max[0];
    a.meta[1] = (4ll * n <= 4000) ? 1.0f : 0.0f;
  }
}

__global__ __launch_bounds__(64) void mask_kernel(NmsArgs a, int fast) {
  __shared__ float sx1[64], sy1[64], sx2[64], sy2[64], sar[64];
  __shared__ int scl[64];
  const int n = clamp_count(a.count, a.cap);
  if (fast && n <= NMS_FAST_N) return;
  const int nblk = (n + 63) >> 6;
  const bool trick = a.meta[1] != 0.0f;
  const float off1 = a.meta[0] + 1.0f;   // max_coordinate + 1
  const int lane = threadIdx.x;
  for (long long p = blockIdx.x; p < (long long)nblk * nblk; p += gridDim.x) {
    int rb = (int)(p / nblk), cb = (int)(p - (long long)rb * nblk);
    if (cb < rb) continue;
    __syncthreads();
    {
      int j = cb * 64 + lane;
      if (j < n) {
        Cand c = a.sorted[j];
        float o = trick ? (float)c.cls * off1 : 0.0f;
        float x1 = trick ? c.x1 + o : c.x1, y1 = trick ? c.y1 + o : c.y1;
        float x2 = trick ? c.x2 + o : c.x2, y2 = trick ? c.y2 + o : c.y2;
        sx1[lane] = x1; sy1[lane] = y1; sx2[lane] = x2; sy2[lane] = y2;
        sar[lane] = (x2 - x1) * (y2 - y1);
        scl[lane] = c.cls;
      }
    }
    __syncthreads();
    int i = rb * 64 + lane;
    if (i < n) {
      Cand c = a.sorted[i];
      float o = trick ? (float)c.cls * off1 : 0.0f;
      float ix1 = trick ? c.x1 + o : c.x1, iy1 = trick ? c.y1 + o : c.y1;
      float ix2 = trick ? c.x2 + o : c.x2, iy2 = trick ? c.y2 + o : c.y2;
      float iarea = (ix2 - ix1) * (iy2 - iy1);
      int lim = n - cb * 64 < 64 ? n - cb * 64 : 64;
      unsigned long long bits = 0ull;
      for (int q = 0; q < lim; ++q) {
        int j = cb * 64 + q;
        if (j <= i) continue;
        if (!trick && scl[q] != c.cls) continue;
        float xx1 = fmaxf(ix1, sx1[q]);
        float yy1 = fmaxf(iy1, sy1[q]);
        float xx2 = fminf(ix2, sx2[q]);
        float yy2 = fminf(iy2, sy2[q]);
        float w = fmaxf(0.0f, xx2 - xx1);
        float h = fmaxf(0.0f, yy2 - yy1);
        float inter = w * h;
        float ovr = inter / (iarea + sar[q] - inter);
        if (ovr > a.iou) bits |= (1ull << q);
      }
      a.mask[(long long)i * nblk + cb] = bits;
    }
  }
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int src) {
  unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xffffffffull), src);
  unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

// single wave: greedy pass over score-ordered boxes
__global__ __launch_bounds__(64) void reduce_kernel(NmsArgs a, int max_words, int fast) {
  OMNI_DYN_LDS(__attribute__((aligned(16))) unsigned long long, remv);
  const int n = clamp_count(a.count, a.cap);
  if (fast && n <= NMS_FAST_N) return;
  const int nblk = (n + 63) >> 6;
  const int lane = threadIdx.x;
  for (int w = lane; w < nblk && w < max_words; w += 64) remv[w] = 0ull;
  __syncthreads();
  int kept_total = 0;
  for (int blk = 0; blk < nblk && kept_total < a.max_det; ++blk) {
    int i = blk * 64 + lane;
    unsigned long long diag = (i < n) ? a.mask[(long long)i * nblk + blk] : 0ull;
    unsigned long long cur = remv[blk];
    int valid = n - blk * 64 < 64 ? n - blk * 64 : 64;
    unsigned long long keptbits = 0ull;
    for (int t = 0; t < valid; ++t) {
      unsigned long long row = readlane64(diag, t);
      if (!((cur >> t) & 1ull)) { keptbits |= (1ull << t); cur |= row; }
    }
    // emit kept boxes of this block in order
    bool mine = (keptbits >> lane) & 1ull;
    int before = __popcll(keptbits & ((1ull << lane) - 1ull));
    int oidx = kept_total + before;
    if (mine && oidx < a.max_det) {
      Cand c = a.sorted[i];
      float fw = (float)a.img_w, fh = (float)a.img_h;
      a.out_boxes[oidx * 4 + 0] = fminf(fmaxf(c.x1, 0.0f), fw);
      a.out_boxes[oidx * 4 + 1] = fminf(fmaxf(c.y1, 0.0f), fh);
      a.out_boxes[oidx * 4 + 2] = fminf(fmaxf(c.x2, 0.0f), fw);
      a.out_boxes[oidx * 4 + 3] = fminf(fmaxf(c.y2, 0.0f), fh);
      a.out_scores[oidx] = c.score;
      a.out_cls[oidx] = c.cls;
    }
    kept_total += __popcll(keptbits);
    // fold the kept rows into the suppression vector of later blocks
    for (int w = blk + 1 + lane; w < nblk; w += 64) {
      unsigned long long acc = remv[w];
      unsigned long long kb = keptbits;
      while (kb) {
        int t = __ffsll((long long)kb) - 1;
        kb &= kb - 1;
        acc |= a.mask[(long long)(blk * 64 + t) * nblk + w];
      }
      remv[w] = acc;
    }
    __syncthreads();
  }
  if (lane == 0) *a.out_count = kept_total < a.max_det ? kept_total : a.max_det;
}

// ---------------------------------------------------------------------------------------------
// One workgroup per frame: sort + greedy NMS of up to NMS_FAST_N candidates without leaving the CU.
__device__ __forceinline__ unsigned f32_bits(float v) { unsigned u; __builtin_memcpy(&u, &v, 4); return u; }

__device__ __forceinline__ unsigned long long or_reduce_wave(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffull), o);
    unsigned hi = __shfl_xor((unsigned)(v >> 32), o);
    v |= ((unsigned long long)hi << 32) | lo;
  }
  return v;
}

__global__ __launch_bounds__(1024) void nms_block_kernel(NmsArgs a) {
  __shared__ unsigned long long key[NMS_FAST_N];           // 16 KB   sort keys; afterwards: sorted position -> (score bits, slot)
  __shared__ float bx1[NMS_FAST_N], by1[NMS_FAST_N], bx2[NMS_FAST_N], by2[NMS_FAST_N], bar[NMS_FAST_N];   // 40 KB offset boxes, areas
  __shared__ short bcl[NMS_FAST_N];                        // 4 KB    class ids
  __shared__ unsigned char dead[NMS_FAST_N];               // 2 KB    suppressed flags (one owner thread per candidate: plain stores)
  __shared__ float red[16];
  __shared__ unsigned long long s_kept;
  __shared__ int s_total;
  const int f = blockIdx.x;
  const Cand* cand = a.cand + (long long)f * a.cap;
  Cand* sorted = a.sorted + (long long)f * (a.cap + 1);
  const int n = clamp_count(a.count + f, a.cap);
  if (n > NMS_FAST_N) return;                               // the tiled path (rank / mask / reduce kernels) handles this frame
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* out_boxes = a.out_boxes + (long long)f * a.max_det * 4;
  float* out_scores = a.out_scores + (long long)f * a.max_det;
  int* out_cls = a.out_cls + (long long)f * a.max_det;
  if (n == 0) { if (tid == 0) a.out_count[f] = 0; return; }
  // ---- keys + max coordinate (torchvision.ops.batched_nms: offsets = idxs * (boxes.max() + 1) when numel <= 4000)
  int N = 64; while (N < n) N <<= 1;
  float m = -INFINITY;
  for (int i = tid; i < N; i += 1024) {
    unsigned long long k = ~0ull;
    if (i < n) {
      Cand c = cand[i];
      m = fmaxf(m, fmaxf(fmaxf(c.x1, c.y1), fmaxf(c.x2, c.y2)));
      // anchor is caller-supplied through the ABI (OMNI_OP_NMS p0): saturated into its 21 key bits so that a value outside
      // [0, 2^21) — which OMNI_OP_DETECT_DECODE never produces — cannot spill into the score bits; such candidates order by slot
      const unsigned an = min((unsigned)c.anchor, (1u << NMS_FAST_ANCHOR_BITS) - 1u);
      k = ((unsigned long long)(~f32_bits(c.score)) << 32) | ((unsigned long long)an << 11) | (unsigned)i;
    }
    key[i] = k;
    dead[i] = 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  float mx = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
  const bool trick = 4ll * n <= 4000;
  const float off1 = mx + 1.0f;
  // ---- bitonic sort, ascending keys = score descending, then anchor ascending (scores are positive: their bit patterns order them)
  for (int k2 = 2; k2 <= N; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (N >> 1); t += 1024) {
        int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        int hi = lo | j;
        unsigned long long x = key[lo], y = key[hi];
        bool up = (lo & k2) == 0;
        if ((x > y) == up) { key[lo] = y; key[hi] = x; }
      }
      __syncthreads();
    }
  }
  // ---- sorted records -> LDS (offset boxes exactly as the tiled path computes them) and the sorted scratch
  for (int i = tid; i < n; i += 1024) {
    Cand c = cand[(int)(key[i] & 2047ull)];
    float o = trick ? (float)c.cls * off1 : 0.0f;
    float x1 = trick ? c.x1 + o : c.x1, y1 = trick ? c.y1 + o : c.y1;
    float x2 = trick ? c.x2 + o : c.x2, y2 = trick ? c.y2 + o : c.y2;
    bx1[i] = x1; by1[i] = y1; bx2[i] = x2; by2[i] = y2;
    bar[i] = (x2 - x1) * (y2 - y1);
    bcl[i] = (short)c.cls;
    sorted[i] = c;
  }
  if (tid == 0) s_total = 0;
  __syncthreads();
  const int nblk = (n + 63) >> 6;
  const float thr = a.iou;
  // ---- every block's own 64x64 mask, all 16 waves at once (wave w: blocks w, w + 16, ...): lane = row (the earlier, possibly kept box),
  //      bit q = a later box of the same block it would suppress.  The key array is free now (the gather above was its last reader).
  unsigned long long* diag = key;
  for (int blk = wave; blk < nblk; blk += 16) {
    const int i = blk * 64 + lane;
    const int valid = n - blk * 64 < 64 ? n - blk * 64 : 64;
    unsigned long long bits = 0ull;
    const int ii = i < n ? i : n - 1;
    const float ix1 = bx1[ii], iy1 = by1[ii], ix2 = bx2[ii], iy2 = by2[ii], iarea = bar[ii];
    const int icl = bcl[ii];
#pragma unroll 4
    for (int q = 1; q < valid; ++q) {                         // uniform trip count: every lane reads the SAME column q (LDS broadcast)
      const int j = blk * 64 + q;
      float xx1 = fmaxf(ix1, bx1[j]);
      float yy1 = fmaxf(iy1, by1[j]);
      float xx2 = fminf(ix2, bx2[j]);
      float yy2 = fminf(iy2, by2[j]);
      float w = fmaxf(0.0f, xx2 - xx1);
      float h = fmaxf(0.0f, yy2 - yy1);
      float inter = w * h;
      float ovr = inter / (iarea + bar[j] - inter);
      if (q > lane && ovr > thr && (trick || bcl[j] == icl)) bits |= (1ull << q);
    }
    if (i < n) diag[i] = bits;
  }
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    if (wave == 0) {
      const int i = blk * 64 + lane;
      const int valid = n - blk * 64 < 64 ? n - blk * 64 : 64;
      const unsigned long long bits = i < n ? diag[i] : 0ull;
      const int total = s_total;                              // read by every lane BEFORE the wave-collective steps below; lane 0 updates it after them
      unsigned long long cur = or_reduce_wave((i < n && dead[i]) ? (1ull << lane) : 0ull);
      unsigned long long keptbits = 0ull;
      for (int t = 0; t < valid; ++t) {
        unsigned long long row = readlane64(bits, t);
        if (!((cur >> t) & 1ull)) { keptbits |= (1ull << t); cur |= row; }
      }
      const bool mine = (keptbits >> lane) & 1ull;
      const int oidx = total + __popcll(keptbits & ((1ull << lane) - 1ull));
      if (mine && oidx < a.max_det) {
        Cand c = sorted[i];                                   // written by this workgroup above (same wave order: visible after the barrier)
        float fw = (float)a.img_w, fh = (float)a.img_h;
        out_boxes[oidx * 4 + 0] = fminf(fmaxf(c.x1, 0.0f), fw);
        out_boxes[oidx * 4 + 1] = fminf(fmaxf(c.y1, 0.0f), fh);
        out_boxes[oidx * 4 + 2] = fminf(fmaxf(c.x2, 0.0f), fw);
        out_boxes[oidx * 4 + 3] = fminf(fmaxf(c.y2, 0.0f), fh);
        out_scores[oidx] = c.score;
        out_cls[oidx] = c.cls;
      }
      if (lane == 0) { s_kept = keptbits; s_total = total + __popcll(keptbits); }
    }
    __syncthreads();
    if (s_total >= a.max_det) break;                          // the first max_det keeps are final: nothing later can change them
    // every later candidate against this block's kept boxes (the kept box is the ROW operand, as in the mask above)
    const unsigned long long kept = s_kept;
    if (kept) {
      for (int j = (blk + 1) * 64 + tid; j < n; j += 1024) {
        if (dead[j]) continue;
        const float jx1 = bx1[j], jy1 = by1[j], jx2 = bx2[j], jy2 = by2[j], jarea = bar[j];
        const int jcl = bcl[j];
        unsigned long long kb = kept;
        while (kb) {
          const int t = __ffsll((long long)kb) - 1;
          kb &= kb - 1;
          const int i = blk * 64 + t;
          if (!trick && bcl[i] != jcl) continue;
          float xx1 = fmaxf(bx1[i], jx1);
          float yy1 = fmaxf(by1[i], jy1);
          float xx2 = fminf(bx2[i], jx2);
          float yy2 = fminf(by2[i], jy2);
          float w = fmaxf(0.0f, xx2 - xx1);
          float h = fmaxf(0.0f, yy2 - yy1);
          float inter = w * h;
          float ovr = inter / (bar[i] + jarea - inter);
          if (ovr > thr) { dead[j] = 1; break; }
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) a.out_count[f] = s_total < a.max_det ? s_total : a.max_det;
}

__global__ void zero_count_kernel(int* count) { if (threadIdx.x == 0) *count = 0; }

}  // namespace

int omni_launch_detect_decode(const omni_op_t* op, hipStream_t s) {
  DecodeArgs a{};
  a.nc = op->i[0];
  int TH = op->i[1], TW = op->i[2];
  OMNI_REQUIRE(a.nc > 0 && TH > 0 && TW > 0 && TH % 32 == 0 && TW % 32 == 0, "detect_decode: bad nc/size");
  int tot = 0;
  for (int l = 0; l < 3; ++l) {
    a.cls[l] = op->p[l]; a.box[l] = op->p[3 + l];
    OMNI_REQUIRE(a.cls[l] && a.box[l], "detect_decode: null head pointer");
    a.ldc[l] = op->i[3 + l]; a.ldb[l] = op->i[6 + l];
    a.coffc[l] = op->i[13 + l]; a.coffb[l] = op->i[16 + l];
    a.hs[l] = TH / (8 << l); a.ws[l] = TW / (8 << l);
    a.astart[l] = tot; tot += a.hs[l] * a.ws[l];
  }
  a.astart[3] = tot;
  a.cap = op->i[9]; a.pad_left = op->i[10]; a.pad_top = op->i[11]; a.dist_reduced = op->i[12];
  a.conf = op->f[0]; a.scale = op->f[1];
  a.cand = (Cand*)op->p[6]; a.count = (int*)op->p[7];
  OMNI_REQUIRE(a.cand && a.count && a.cap > 0, "detect_decode: null output");
  OMNI_REQUIRE(a.scale > 0.0f, "detect_decode: bad scale");
  // the candidate counter is reset by a KERNEL, not by hipMemsetAsync: a captured plan then consists of kernel nodes only (a memset
  // node in the middle of the detector graph was the one non-kernel node of the graph whose second replay stalled in round 2,
  // profiles/r2_notes.md)
  hipLaunchKernelGGL(zero_count_kernel, dim3(1), dim3(64), 0, s, a.count);
  dim3 grid((tot + 255) / 256);
  if (op->dtype == OMNI_F32) hipLaunchKernelGGL(decode_kernel<float>, grid, dim3(256), 0, s, a);
  else if (op->dtype == OMNI_F16) hipLaunchKernelGGL(decode_kernel<half_t>, grid, dim3(256), 0, s, a);
  else OMNI_REQUIRE(false, "detect_decode: bad dtype");
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

int omni_launch_nms(const omni_op_t* op, hipStream_t s) {
  NmsArgs a{};
  a.cand = (const Cand*)op->p[0]; a.count = (const int*)op->p[1];
  a.sorted = (Cand*)op->p[2]; a.mask = (unsigned long long*)op->p[3];
  a.out_boxes = (float*)op->p[4]; a.out_scores = (float*)op->p[5];
  a.out_cls = (int*)op->p[6]; a.out_count = (int*)op->p[7];
  a.cap = op->i[0]; a.max_det = op->i[1]; a.img_w = op->i[2]; a.img_h = op->i[3];
  a.iou = op->f[0];
  a.frames = op->i[4] > 0 ? op->i[4] : 1;
  OMNI_REQUIRE(a.cand && a.count && a.sorted && a.mask && a.out_boxes && a.out_scores && a.out_cls && a.out_count,
               "nms: null pointer");
  OMNI_REQUIRE(a.cap > 0 && a.max_det > 0, "nms: bad cap/max_det");
  int max_words = (a.cap + 63) / 64;
  OMNI_REQUIRE((size_t)max_words * 8 <= 160 * 1024, "nms: cap too large for the LDS suppression vector");
  // up to NMS_FAST_N candidates: one workgroup per frame does everything (sort, masks, greedy pass) out of LDS.  The candidate count is
  // known on the device only, so the tiled kernels are launched as well and return at once for the frames the fast kernel took (and
  // the fast kernel returns for the frames it cannot hold): an empty launch inside a graph costs ~2 us, a host round trip ~20.
  const int fast = (a.cap < (1 << NMS_FAST_ANCHOR_BITS)) && op->i[5] != 1 ? 1 : 0;          // i5 = 1: force the tiled path (tests)
  if (fast) hipLaunchKernelGGL(nms_block_kernel, dim3(a.frames), dim3(1024), 0, s, a);
  if (a.cap > NMS_FAST_N || !fast) {
    for (int f = 0; f < a.frames; ++f) {
      NmsArgs b = a;
      b.cand = a.cand + (long long)f * a.cap; b.count = a.count + f; b.sorted = a.sorted + (long long)f * (a.cap + 1);
      b.out_boxes = a.out_boxes + (long long)f * a.max_det * 4; b.out_scores = a.out_scores + (long long)f * a.max_det;
      b.out_cls = a.out_cls + (long long)f * a.max_det; b.out_count = a.out_count + f;
      // sorted scratch holds cap records + 1 spare record used for {max_coord, trick flag}
      b.meta = reinterpret_cast<float*>(b.sorted + b.cap);
      int rank_blocks = (b.cap + 255) / 256; if (rank_blocks > 1024) rank_blocks = 1024;
      hipLaunchKernelGGL(nms_prep_kernel, dim3(1), dim3(256), 0, s, b, fast);
      hipLaunchKernelGGL(rank_kernel, dim3(rank_blocks), dim3(256), 0, s, b, fast);
      long long pairs = (long long)max_words * max_words;
      int mask_blocks = pairs < 4096 ? (int)pairs : 4096;
      hipLaunchKernelGGL(mask_kernel, dim3(mask_blocks), dim3(64), 0, s, b, fast);
      hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(64), (size_t)max_words * 8, s, b, max_words, fast);
    }
  }
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}
