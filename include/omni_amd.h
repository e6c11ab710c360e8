/*
 * omni_amd.h — C ABI of libomni_amd.so, the MI355X (gfx950) screen-parsing hot path.
 *
 * The reference (microsoft/OmniParser) has no FFI: its hot path is Python calling
 * third-party kernels (TorchScript YOLOv9-E, torchvision::nms, transformers
 * Florence-2).  This header is the boundary a maintainer binds with ctypes from
 * the reference's own adapter classes (INTEGRATION.md shows the stubs).  Each
 * entry point cites the reference call it replaces (ref: = /root/reference,
 * hf: = site-packages/transformers).
 *
 * Conventions
 *   - plain pointers + sizes; every pointer named d_* is DEVICE memory owned by
 *     the caller (torch tensors in the Python host), h_* is host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     call is asynchronous on that stream unless documented otherwise.
 *   - return 0 on success, negative OMNI_E_* on failure; omni_last_error()
 *     returns a thread-local message.
 *   - activations are NHWC (tokens x channels for the captioner), dtype
 *     OMNI_F32 (f32 storage, accumulation and statistics; GEMMs multiply on the f16
 *     matrix cores with both operands split into two f16 halves and THREE products
 *     per MAC — measured < 2e-6 relative to f64, the parity mode; short-K layers use
 *     the exact f32 MFMA) or OMNI_F16 (f16 storage and MFMA, f32 accumulate).
 */
#ifndef OMNI_AMD_H
#define OMNI_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): omni_stream_create / omni_stream_destroy / omni_plan_run_split and omni_debug_host_op are gone; OMNI_OP_ATTN_ROWS i17,
 * OMNI_OP_CHAN_ATTN i7 and OMNI_OP_CONV p6 are ignored (i22 / i23 = tile / split-K override); OMNI_OP_NMS sorts candidates inside the op.  A host compiled against
 * version 1 must be rebuilt: omni_abi_version() is what it checks at load time (omniparser_amd/_lib.py does). */
/* 3 (round 6): + omni_overflow_count; plan bundles carry the ABI version they were written under (omni_model_load rejects others). */
#define OMNI_ABI_VERSION 3

enum { OMNI_F32 = 0, OMNI_F16 = 1 };
enum { OMNI_ACT_NONE = 0, OMNI_ACT_SILU = 1, OMNI_ACT_GELU = 2 };
enum {
  OMNI_OK = 0,
  OMNI_E_ARG = -1,     /* bad argument / unsupported shape */
  OMNI_E_HIP = -2,     /* HIP runtime error */
  OMNI_E_NODEV = -3    /* no gfx950 device */
};

const char* omni_last_error(void);
int omni_abi_version(void);
/* number of visible HIP devices, <0 on error. */
int omni_device_count(void);
/* Range guard of the split-f16 formats (ABI 3).  OMNI_F32 plans keep the hi half of every GEMM operand in f16, so an activation
 * with |x| > 65504 cannot be represented (format B clamps it, format A loses it); the fp32 reference has no such limit
 * (ref:util/utils.py:66 loads the captioner in float32 on the CPU).  Every kernel that produces such an operand counts the
 * threads that saw a value beyond the range; *count = that number since the last reset (0 on every tensor of a healthy model).
 * Synchronous (a device-to-host copy of a few words): call it where the host already waits for results.  reset != 0 zeroes the
 * counters. */
int omni_overflow_count(int reset, unsigned long long* count);

/* ------------------------------------------------------------------------ *
 * Generic op descriptor: the plan executor and the single-op entry point
 * share it.  Slot meaning per kind is documented at each OMNI_OP_* below.
 * ------------------------------------------------------------------------ */
typedef struct omni_op {
  int32_t kind;      /* OMNI_OP_* */
  int32_t dtype;     /* OMNI_F32 / OMNI_F16: activation + weight element type */
  void*   p[8];      /* device pointers */
  int32_t i[32];     /* integer parameters */
  float   f[8];      /* float parameters */
} omni_op_t;

enum {
  /* Implicit-GEMM convolution / linear layer on MFMA, fused bias + act + residual.
   * Replaces the Conv2d(+BN)+SiLU stack inside the TorchScript detector
   * (ref:util/yolov9.py:121) and nn.Linear/Conv2d in the captioner
   * (hf:models/florence2/modeling_florence2.py, hf:models/bart/modeling_bart.py).
   *  p0 x [B,H,W,ldi]   p1 w [Cout][KH*KW*Cin] (k = (r*KW+s)*Cin + c)   p2 bias f32[Cout] or NULL
   *  p3 residual [M,ldr] or NULL   p4 y [B,Ho,Wo,ldo]
   *  i0 B i1 H i2 W i3 Cin i4 ldi i5 in_coff i6 KH i7 KW i8 stride i9 pad i10 Ho i11 Wo
   *  i12 Cout i13 ldo i14 out_coff i15 act i16 ldr i17 res_coff
   *  p5 optional split-K workspace (f32), i19 its size in KiB   f0 output scale (0 => 1)
   *  i20 = 1: split-f16 mode (f32 activations; w = [Cout][K/16][16 hi | 16 lo] f16 halves with w = hi + lo*2^-11;
   *           three f16 MFMAs per block give f32-class accuracy at the f16 matrix rate; needs Cin % 32 == 0)
   *  i20 = 2: pre-split LDS-DMA GEMM (csrc/gemm_dma.hip; pointwise layers only, K % 32 == 0, Cout % 128 == 0):
   *           BOTH operands are "format B" f16 pairs — a 16-channel group is 64 bytes, 16 hi halves then 16 lo halves,
   *           value = hi + lo (4 bytes per element, same strides as f32) — x written that way by its producer
   *           (OMNI_OP_LAYERNORM i6, this op's i21, OMNI_OP_SPLIT_CONVERT), w = split(W * 2^k) with f1 = 2^-k;
   *           i21 = 1: write y in format B as well (bias + activation applied first; no residual)
   *  i22 / i23 (i20 = 1 only; 0 = the launcher's heuristic): output tile (1 = 64x64, 2 = 128x64, 3 = 128x128) and split-K count
   *           (1 = no split, no reduce launch) — the per-shape choices of a tuning table; the sums differ only in the order of the K partials
   *  i24 = n > 0 (i20 = 1 only; round 6): p6 = int32[n] ARRIVAL COUNTERS, all zero before the first launch that uses them.  A split-K
   *           launch with at most n output tiles then combines its partials INSIDE the conv launch (each split publishes its tile
   *           write-through and draws a ticket; the last arriver sums in split order, applies bias / act / residual and zeroes the
   *           counter) instead of a second reduce launch: same sums in the same order, one launch less per conv.  Launches that share
   *           counters (or the workspace p5) must not run concurrently: one plan = one stream.  i24 = 0: the reduce launch
   *  i25 = 1 (i20 = 1 only; round 6): ROW-PATCH mode for a k x k convolution over ldi = 4 stored channels (k <= 8; the captioner's
   *           first patch embedding): pass it as a k x 1 convolution over 8 consecutive pixels — i6 KH = k, i7 KW = 1, i3 Cin = 32,
   *           i4 ldi = 4, i5 = 0, i8 / i9 stride / pad of BOTH axes, w = [Cout][k][8 pixels][4] with zero weights for pixel >= k (split
   *           like any i20 = 1 weight).  Pixels outside the image read as zero one by one; nothing is read beyond a row */
  OMNI_OP_CONV = 1,
  /* avg_pool2d(k=2,s=1,p=0) (ADown, ref blob T1).  p0 x, p4 y.
   *  i0 B i1 H i2 W i3 C i4 ldi i5 in_coff i13 ldo i14 out_coff (Ho=H-1, Wo=W-1) */
  OMNI_OP_AVGPOOL2 = 2,
  /* max_pool2d(k,s,p) with -inf padding (ADown k3s2p1, SPP k5s1p2).
   *  p0 x p4 y; i0 B i1 H i2 W i3 C i4 ldi i5 in_coff i6 k i8 stride i9 pad i10 Ho i11 Wo i13 ldo i14 out_coff */
  OMNI_OP_MAXPOOL = 3,
  /* nearest-neighbour resize (F.interpolate mode='nearest') of a channel slice,
   * either overwriting or accumulating into y (Upsample, CBFuse).
   *  p0 x p4 y; i0 B i1 H i2 W i3 C i4 ldi i5 in_coff i10 Ho i11 Wo i13 ldo i14 out_coff i18 accumulate
   *  i17 = n > 1 (CBFuse in one launch): y = ((r(x0) + r(x1)) + ...) over n <= 5 sources of C channels in order, r = nearest resize to
   *  Ho x Wo; source 0 as above, sources 1..4 = p1, p2, p3, p5 with (H, W, ld, coff) in i19-22, i23-26, i27-30, (i7, i12, i15, i16);
   *  the same partial sums, rounded the same way, as n launches with i18 = 1 */
  OMNI_OP_RESIZE_NEAREST = 4,
  /* Pillow-exact separable resample (LANCZOS/BICUBIC, 8bpc fixed point) +
   * letterbox + /255 -> network input.  Replaces ref:util/yolov9.py:73-87.
   *  p0 img u8 [H,W,3]  p1 tmp u8 [H,Wr,3]  p2 xbounds i32[Wr*2] p3 xcoef i32[Wr*kx]
   *  p5 ybounds i32[Hr*2] p6 ycoef i32[Hr*ky]  p4 y [1,TH,TW,ldo] (channels >=3 zeroed)
   *  i0 H i1 W i2 Hr i3 Wr i4 kx i5 ky i6 TH i7 TW i8 pad_left i9 pad_top i13 ldo
   *  i10 need_h i11 need_v (0 => that pass is a copy, Pillow's same-size shortcut) i12 batch index */
  OMNI_OP_LETTERBOX = 5,
  /* DFL expectation + anchor decode + sigmoid + max-class + threshold +
   * un-letterbox + compaction.  Replaces ref:util/yolov9.py:89-129.
   *  p0..p2 cls logits per stride [H_s*W_s, ldc_s]; p3..p5 box logits [H_s*W_s, ldb_s] (4*16 DFL bins, or 4 if i12==1)
   *  p6 cand (omni_cand_t[cap])  p7 count i32[1]
   *  i0 nc i1 TH i2 TW (network input size) i3..i5 ldc_s i6..i8 ldb_s i9 cap i10 pad_left i11 pad_top
   *  i12 dist_is_reduced i13..i15 cls channel offsets i16..i18 box channel offsets i19 batch index
   *  f0 conf f1 scale */
  OMNI_OP_DETECT_DECODE = 6,
  /* class-aware greedy NMS identical to torchvision.ops.batched_nms on CPU
   * (ref:util/yolov9.py:131) + [:max_det] + clamp (:134-135).
   *  p0 cand p1 count p2 sorted cand scratch (cap) p3 mask scratch u64[cap*ceil(cap/64)]
   *  p4 out boxes f32[max_det,4] p5 out scores f32[max_det] p6 out cls i32[max_det] p7 out count i32[1]
   *  i0 cap i1 max_det i2 img_w i3 img_h ; f0 iou
   *  i4 frames (0 = 1): frame f reads cand + f*cap records, count[f], sorted + f*(cap+1) records, writes out_* + f*max_det, out_count[f]
   *  (the mask scratch is shared: frames that need it run one after another); i5 = 1: force the tiled kernels (tests).
   *  Frames with <= 2048 candidates are sorted and suppressed by one workgroup out of LDS; the result is the same list either way.
   *  omni_cand_t.anchor (the tie-break of equal scores = torch's stable sort over anchor order) must lie in [0, 2^21) on that path —
   *  true of everything OMNI_OP_DETECT_DECODE writes; values outside are saturated (never mis-sorted against the score) and tie-break
   *  by slot.  A caller with larger anchor ids sets i5 = 1. */
  OMNI_OP_NMS = 7,
  /* x + depthwise3x3(x) + bias (DaViT conv1/conv2, hf:models/florence2/modeling_florence2.py:432-436).
   *  p0 x [B,H,W,C] p1 w [3][3][C] p2 bias f32[C] p4 y; i0 B i1 H i2 W i3 C */
  OMNI_OP_DWCONV3 = 8,
  /* nn.LayerNorm over C (<= 1024) of x (+ add[row % period]) (hf florence2 :154-574, bart :272-341).
   *  p0 x [rows,C] p1 add [period,C] or NULL p2 gamma f32 p3 beta f32 p4 y; i0*i1 rows i3 C i5 period; f0 eps
   *  i6 output mode (f32 plans, C % 16 == 0): 0 = f32 y, 1 = y in format B (see OMNI_OP_CONV i20 = 2), 2 = f32 y AND format B p5 */
  OMNI_OP_LAYERNORM = 9,
  /* softmax(q k^T * scale) v, one query row per thread; mode 0 plain MHA (bart :143-257), mode 1 DaViT
   * 12x12 window attention incl. the unmasked zero-padded window tokens (florence2 :338-398).
   *  p0 q p1 k p2 v (token matrices) p4 o p5 kbias f32 p6 vbias f32 (window padding rows)
   *  i0 ldq i1 ldk i2 ldv i3 ldo i4 qoff i5 koff i6 voff i7 ooff i8 heads i9 nq i10 nk i11 groups
   *  i12 mode i13 H i14 W i15 head_dim (32|64); f0 scale
   *  i16 = 1 (f32 plans, MFMA kernels): write o in format B (see OMNI_OP_CONV i20 = 2) for the LDS-DMA GEMM that follows
   *  f32 plans, MFMA kernels (mode 1 with head_dim 32, mode 0 with head_dim 64): row pitches and channel offsets % 4 == 0 */
  OMNI_OP_ATTN_ROWS = 10,
  /* DaViT grouped channel attention (florence2 :223-259): p0 qkv [B*N,3C] p4 o [B*N,C] p5 ws f32[B*G*chunks*1024]
   *  i0 B i1 N i3 C i4 G i5 chunk_tokens i6 = 1: o in format B (f32 plans); f0 scale (0 => N^-0.5) */
  OMNI_OP_CHAN_ATTN = 11,
  /* projector input (florence2 :568-590): y[b] = [mean_n(x+pos+t) ; x+pos+t]; p0 x [B,N,C] p1 pos2d f32[N,C] p2 temporal f32[C] p4 y [B,N+1,C]
   *  i0 B i1 N i3 C */
  OMNI_OP_PROJ_PREP = 12,
  /* encoder input = [image features ; prompt embeddings] (florence2 :933-960): p0 img [B,n_img,C] p1 txt [n_txt,C] p4 y
   *  i0 B i1 n_img i2 n_txt i3 C */
  OMNI_OP_ASSEMBLE = 13,
  /* decoder token embedding + learned position (bart :80-98): y[b] = table[ids[b][step]]*scale + pos[step+off]
   *  p0 table p1 pos p2 ids i32[B,T] p4 y [B,C] p6 step i32*; i0 B i3 C i4 T i5 pos offset; f0 scale */
  OMNI_OP_EMBED_STEP = 14,
  /* single-query attention, head_dim 64: self (append k/v at `step` to the cache, attend 0..step) or cross
   * (nk_fixed keys).  p0 q p1 knew p2 vnew p3 kcache [B,cap,C] p5 vcache p4 o [B,ldo] p6 step
   *  i0 ldq i1 qoff i2 ldn i3 koff i4 voff i5 ldo i6 heads i7 nk_fixed i8 cap i9 C i10 B i11 cache row stride (0 => C); f0 scale */
  OMNI_OP_ATTN_DECODE = 15,
  /* greedy decoding step (hf:generation/utils.py:2783-2937 + logits_process NoRepeatNGram/ForcedBOS/ForcedEOS):
   *  p0 logits [B,ldl] p1 final_logits_bias f32 or NULL p2 ids i32[B,T] p3 finished i32[B] p6 step i32*
   *  i0 B i1 vocab i2 ldl i3 T i4 max_new_tokens i5 no_repeat_ngram i6 bos i7 eos i8 pad i9 forced_bos(-1)
   *  i10 forced_eos(-1) i11 increment step afterwards */
  OMNI_OP_GREEDY_STEP = 16,
  /* crop -> cv2.resize 64x64 INTER_LINEAR -> [Pillow BICUBIC to RxR] -> rescale, normalise
   * (ref:util/utils.py:97-105,120-123 + hf CLIP image processor).
   *  p0 img u8 [H,W,3] p1 boxes i32[n,4] (x0,y0,x1,y1 px) p2 c64 u8[n,64,64,3] p3 tmp u8[n,64,R,3]
   *  p4 y [n,R,R,ldo] p5 bounds i32[R,2] p6 coef i32[R,ksize] p7 lut f32[256]
   *  i0 n i1 H i2 W i3 R i4 ksize i13 ldo; f0..2 mean f3..5 std */
  OMNI_OP_CROP_RESIZE = 17,
  /* fused x1 = x + depthwise3x3(x) + bias ; h = LayerNorm(x1) (DaViT half-block prologue, hf florence2 :432-441).
   *  p0 x [B,H,W,C] p1 w [3][3][C] p2 bias f32[C] p3 h [B,H,W,C] p4 x1 [B,H,W,C] p5 gamma f32 p6 beta f32
   *  i0 B i1 H i2 W i3 C (<= 1024) i6 = 1: h in format B (f32 plans); f0 eps */
  OMNI_OP_DWCONV3_LN = 18,
  /* f32 channel slice of a token matrix -> format B (OMNI_OP_CONV i20 = 2), in place when p0 == p4 and the slices coincide.
   *  p0 x [rows, ldi] p4 y [rows, ldo]; i0*i1 rows i3 C i4 ldi i5 in_coff i13 ldo i14 out_coff (all multiples of 16) */
  OMNI_OP_SPLIT_CONVERT = 19,
  /* Detect -> caption hand-off of ONE screenshot on the device: ref:util/utils.py:432-453 (ratio boxes, int_box_area filter),
   * remove_overlap_new (:241-319, incl. its list.remove quirk), the "content is None last" ordering (:449-451) and the crop
   * rectangles of get_parsed_content_icon (:90-98), in the reference's own mix of f32 / Python-float (f64) arithmetic.
   *  p0 boxes f32[max_det,4] (px, score order; ratios if i5) p1 box count i32* p2 OCR boxes f64[cap,4] (ratios, already
   *  int_box_area-filtered by the host) p3 OCR table i32[2 + 2*cap]: {live OCR count, 0, (equality class, rank in class) per box}
   *  p4 out element table i32[i6,2]: (kind 0 OCR | 1 icon with OCR text | 2 icon to caption, source index) in output order
   *  p5 out crop rectangles i32[max_det,4] (x0,y0,x1,y1 px, caption order) p6 out i32[4]: {elements, crops, index of the
   *  first caption-less element or -1, surviving OCR boxes} p7 out donor bit masks u64[max_det, i4] (OCR boxes whose text an
   *  icon collects)
   *  i0 max_det (<= 512) i1 OCR capacity (<= 1024) i2 W i3 H i4 mask words per icon i5 boxes are ratios i6 element capacity
   *  i7 = 1: the overlap threshold is the f64 with bit pattern i9:i8 (Python passes 0.7 as a double), else f0 */
  OMNI_OP_GLUE = 20,
  /* Set-of-marks overlay rastered in place on the device (ref:util/utils.py:478-483 annotate() -> ref:util/box_annotator.py:86-162;
   * the draw list comes from util/overlay.py::plan_overlay, pinned to the reference's cv2 call sequence): painter's algorithm over
   * a primitive list, in order.  p0 frame u8[H,W,3] (in / out) p1 primitives i32[n,8] = {kind, x0, y0, x1, y1, r | g<<8 | b<<16,
   * a, 0}: kind 0 filled rectangle (x1, y1 inclusive), 1 ring = the pixels of the rectangle that are not in its interior shrunk by
   * a = width on every side, 2 coverage mask (x1, y1 exclusive; a = byte offset of its (y1-y0) x (x1-x0) u8 mask in p2) blended
   * like Pillow's draw_bitmap; p2 masks u8.  i0 H i1 W i2 n */
  OMNI_OP_OVERLAY = 21,
  /* Frame -> PNG file -> base64, all on the device (ref:util/utils.py:485-488: PIL save(format="PNG") + base64.b64encode):
   * signature, IHDR (8-bit RGB), ONE IDAT whose zlib stream holds stored deflate blocks (filter 0 scanlines; 65535-byte blocks),
   * IEND; Adler-32 and the chunk CRC-32 are computed on the device.  File size = H (3 W + 1) + 5 ceil(H (3 W + 1) / 65535) + 63.
   *  p0 frame u8[H,W,3] p1 out PNG bytes p2 scratch u32[2 H + ceil((size - 53) / 4096)] p3 out base64 ASCII (4 ceil(size / 3)
   *  bytes) or NULL.  i0 H i1 W i2 scratch words i3 capacity of p1 in bytes (0 = not checked) */
  OMNI_OP_PNG_PACK = 22,
  /* Same file as OMNI_OP_PNG_PACK with a COMPRESSED zlib stream (size known only on the device): scanlines Up-filtered (row 0:
   * None), the filtered stream cut into 4096-byte units, each unit ONE fixed-Huffman deflate block of literals and run matches
   * (distance 1 / 3) + an empty stored block (byte alignment), or one stored block when that is not smaller — one GPU thread per
   * unit, then offsets, gather, Adler-32, CRC-32, base64 as above.  Worst case = the stored size with 4096-byte blocks.
   *  p0 frame u8[H,W,3] p1 out PNG bytes (capacity i2) p2 scratch: filtered stream u8[H (3 W + 1)] p3 scratch: unit slots
   *  u8[units * 4640] p4 out/scratch meta u32[i3]: {zlib stream bytes, FILE BYTES, base64 bytes, CRC segments, unit sizes..., unit
   *  offsets...} p5 scratch u32[i4] (Adler / CRC partials) p6 out base64 ASCII (4 ceil(capacity / 3) bytes) or NULL
   *  i0 H i1 W i2 capacity of p1 (>= H (3 W + 1) + 5 units + 63) i3 meta words (>= 4 + 2 units) i4 scratch words
   *  (>= 2 H + ceil((capacity - 53) / 4096))
   *  i5 = 1 (round 6, what `OMNI_OVERLAY=device` uses): LZ77 + DYNAMIC Huffman instead — units of 32768 bytes, one GPU lane per
   *  unit: greedy matches against distance 1 / 3 and a 13-bit single-entry hash table of the unit's own 3-byte strings, one dynamic
   *  block per unit (length-limited canonical codes built on the device) + the empty stored block; ~1.1x (desktop screenshots) ...
   *  1.3x (noise-heavy frames) the bytes of Pillow's zlib level 6 (the fixed-Huffman variant: 2.7x).  Needs p3 >= ceil(H (3 W + 1) /
   *  32768) * 33792 bytes and p7 = token scratch u32[ceil(H (3 W + 1) / 32768) * 32768]; same capacities otherwise.  Bytes =
   *  oracle/png_ref.py::deflate_png_lz. */
  OMNI_OP_PNG_DEFLATE = 23,
  /* FFN of a DaViT block as one kernel (hf:models/florence2/modeling_florence2.py Florence2VisionMLP inside the residual of
   * Florence2VisionSpatialBlock / ChannelBlock): y = residual + fc2(GELU(fc1(x))), hidden activations kept in registers
   * (csrc/gemm_dma.hip::mlp_fused_kernel).  f32 plans, C = 128, hidden = 512 (DaViT stage 0 of Florence-2-base).
   *  p0 x [rows, ldi] in format B (see OMNI_OP_CONV i20 = 2)   p1 w1 format B [hidden][C], = split(W1 * 2^k1)   p2 b1 f32[hidden]
   *  p3 residual f32 [rows, ldr]   p4 y f32 [rows, ldo] (may alias the residual)   p6 b2 f32[C]
   *  p5 w2 format B [C][hidden] = split(W2 * 2^k2) with the hidden axis permuted inside every 16-group to
   *     {0,1,2,3, 8,9,10,11, 4,5,6,7, 12,13,14,15} (the order in which the first product's accumulators hold it)
   *  i0*i1 rows i3 C i4 ldi i5 in_coff i12 hidden i13 ldo i14 out_coff i16 ldr i17 res_coff; f1 2^-k1 f2 2^-k2 */
  OMNI_OP_MLP_FUSED = 24,
  OMNI_OP__COUNT
};

/* candidate record shared by DETECT_DECODE and NMS (32 bytes) */
typedef struct omni_cand {
  float x1, y1, x2, y2;   /* un-letterboxed, unclamped image pixels */
  float score;
  int32_t cls;            /* arg-max class id */
  int32_t anchor;         /* flat anchor index over strides 8,16,32 (tie-break key) */
  int32_t pad;
} omni_cand_t;
/* NMS scratch `p2` must hold (cap + 1) records: the spare record carries
 * {max coordinate, coordinate-trick flag} of torchvision.ops.batched_nms. */

/* Launch ONE op on `stream` (unit tests, eager mode).
 * Pointer check (here and in omni_plan_create, wherever a device is present; OMNI_CHECK_PTRS=0 turns it off): every non-NULL p[k]
 * must lie in a device allocation known to the HIP runtime, and the byte range the op touches from it (computed for the conv /
 * GEMM family, pools / resizes incl. the extra CBFuse sources, LayerNorm, depthwise conv, split-convert, the fused FFN, detect-decode
 * and NMS; the FIRST BYTE only for the attention, decode-step, crop, hand-off and PNG kinds) must end inside that allocation —
 * otherwise OMNI_E_ARG with the op index, slot and address in omni_last_error() instead of a GPU memory-access fault (which would
 * abort the process).  What it cannot see: an overrun that stays inside one allocation (with a caching allocator: one segment). */
int omni_op_launch(const omni_op_t* op, void* stream);

/* Plan: an immutable list of ops replayed per inference; optionally captured
 * into a hipGraph (one graph launch per screenshot instead of ~300 kernel launches). */
typedef struct omni_plan omni_plan_t;
int omni_plan_create(const omni_op_t* ops, int n_ops, omni_plan_t** out);
int omni_plan_run(omni_plan_t* plan, void* stream);            /* eager replay */
int omni_plan_capture(omni_plan_t* plan, void* stream);        /* build hipGraphExec */
int omni_plan_replay(omni_plan_t* plan, void* stream);         /* launch captured graph */
int omni_plan_num_ops(const omni_plan_t* plan);
void omni_plan_destroy(omni_plan_t* plan);

/* Pillow `precompute_coeffs` + 8bpc fixed-point normalisation, on the host.
 * filter: 0 = LANCZOS (support 3), 1 = BICUBIC (support 2).
 * Writes h_bounds[2*out_size] (xmin, count) and h_coef[out_size*ksize] (22-bit fixed point).
 * Returns ksize (>0) — call with h_bounds == NULL to query ksize only. */
int omni_resample_coeffs(int in_size, int out_size, int filter, int32_t* h_bounds, int32_t* h_coef);

/* HIP-event timing helper so hosts without torch can time a stream region.
 * Records start, runs `plan` `iters` times (graph replay if captured), records
 * stop, synchronises and returns mean milliseconds per iteration in *ms. */
int omni_plan_time(omni_plan_t* plan, void* stream, int iters, float* ms);

/* One EAGER replay with a HIP event around every op (on `stream`): h_ms[i] = device milliseconds of op i in its real
 * sequence.  bench.py derives `roofline.achieved` and the per-kernel-family split of a step from it; the numbers are
 * directly comparable with `rocprofv3 --kernel-trace --stats` of the same replay.  h_ms holds omni_plan_num_ops floats. */
int omni_plan_profile(omni_plan_t* plan, void* stream, float* h_ms);

/* Host mirror of the GEMM kernels' block -> output-tile permutation (XCD-aware order with an optional N partition
 * over XCD groups; csrc/conv_igemm.hip::tile_of_block).  Test/diagnostic entry point, no device work: for block
 * `bid` of a grid over mtiles x ntiles tiles writes the tile (or -1/-1 for a padding block), the grid size and the
 * N-partition count used.  weight_bytes >= 0: choose the partition like the launcher (L2-residency rule);
 * weight_bytes < 0: use `xcd_n` (1, 2, 4 or 8, must divide ntiles). */
int omni_debug_tile_map(int mtiles, int ntiles, int xcd_n, long long weight_bytes, int bid, int* mt, int* nt, int* grid,
                        int* xcd_n_used);

/* ------------------------------------------------------------------------ *
 * Model-level entry points: the two models of the hot path for hosts
 * without Python.  A model is a PLAN BUNDLE file written by omniparser_amd/bundle.py
 * (export_detector / export_captioner): the op lists above over a fixed set of device
 * buffers, with constants and named I/O tensors.  omni_model_load allocates the
 * buffers, uploads the constants, relocates the ops and captures every plan as a
 * hipGraph on a stream owned by the model.  All calls below are SYNCHRONOUS.
 * ------------------------------------------------------------------------ */
typedef struct omni_model omni_model_t;

int  omni_model_load(const char* bundle_path, omni_model_t** out);
void omni_model_destroy(omni_model_t* model);
/* scalars / named device tensors of a bundle ("batch", "img_w", "capacity", "T", ... ; "img", "out_boxes", "x_in", "ids", ...). */
int  omni_model_int(const omni_model_t* model, const char* name, long long* value);
int  omni_model_tensor(const omni_model_t* model, const char* name, void** d_ptr, long long* nbytes);
/* replay one plan of the bundle ("detect", "encode", "step") and wait for it. */
int  omni_model_run(omni_model_t* model, const char* plan_name);

/* Detector = YOLOv9Detector.predict on one batch (ref:util/yolov9.py:115-136: letterbox, network, decode, threshold,
 * batched_nms[:max_det], clamp).  The bundle fixes image size, network size, thresholds, max_det and batch.
 *  images_rgb  n_images x [img_h, img_w, 3] uint8, host (on_device = 0) or device memory
 *  h_boxes     [n_images, max_det, 4] f32 xyxy pixels; h_scores / h_classes [n_images, max_det] (may be NULL);
 *  h_counts    [n_images] number of valid rows per image. */
int  omni_detector_create(const char* bundle_path, omni_model_t** out);
int  omni_detector_infer(omni_model_t* det, const uint8_t* images_rgb, int n_images, int on_device, float* h_boxes, float* h_scores,
                         int32_t* h_classes, int32_t* h_counts);

/* Captioner = the icon_caption loop of ref:util/utils.py:88-132 on one screenshot: crop each box, cv2.resize 64x64, processor
 * (bicubic to R x R on the 768 path, rescale, normalise), Florence-2 generate(num_beams=1, max_new_tokens) — greedy ids.
 *  image_rgb   [img_h, img_w, 3] uint8, host or device;   h_boxes_px [n, 4] int32 xyxy pixels (any n: micro-batches of the
 *  bundle's capacity);   h_ids [n, T] int32, T = omni_model_int("T") = max_new_tokens + 1; rows are padded with the pad id
 *  after EOS exactly as generate() pads them. */
int  omni_captioner_create(const char* bundle_path, omni_model_t** out);
int  omni_captioner_caption(omni_model_t* cap, const uint8_t* image_rgb, int on_device, int img_h, int img_w, const int32_t* h_boxes_px,
                            int n, int32_t* h_ids);

#ifdef __cplusplus
}
#endif
#endif /* OMNI_AMD_H */
