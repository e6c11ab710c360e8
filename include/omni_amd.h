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
 *     OMNI_F32 (exact-f32 MFMA, parity mode) or OMNI_F16 (f16 MFMA, f32 accumulate).
 */
#ifndef OMNI_AMD_H
#define OMNI_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMNI_ABI_VERSION 1

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
   *  p5 optional split-K workspace (f32), i19 its size in KiB   f0 output scale (0 => 1) */
  OMNI_OP_CONV = 1,
  /* avg_pool2d(k=2,s=1,p=0) (ADown, ref blob T1).  p0 x, p4 y.
   *  i0 B i1 H i2 W i3 C i4 ldi i5 in_coff i13 ldo i14 out_coff (Ho=H-1, Wo=W-1) */
  OMNI_OP_AVGPOOL2 = 2,
  /* max_pool2d(k,s,p) with -inf padding (ADown k3s2p1, SPP k5s1p2).
   *  p0 x p4 y; i0 B i1 H i2 W i3 C i4 ldi i5 in_coff i6 k i8 stride i9 pad i10 Ho i11 Wo i13 ldo i14 out_coff */
  OMNI_OP_MAXPOOL = 3,
  /* nearest-neighbour resize (F.interpolate mode='nearest') of a channel slice,
   * either overwriting or accumulating into y (Upsample, CBFuse).
   *  p0 x p4 y; i0 B i1 H i2 W i3 C i4 ldi i5 in_coff i10 Ho i11 Wo i13 ldo i14 out_coff i18 accumulate */
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
   *  i0 cap i1 max_det i2 img_w i3 img_h ; f0 iou */
  OMNI_OP_NMS = 7,
  /* depthwise 3x3 conv + bias + residual (DaViT conv_pos_enc). */
  OMNI_OP_DWCONV3 = 8,
  OMNI_OP_LAYERNORM = 9,
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

/* Launch ONE op on `stream` (unit tests, eager mode). */
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

#ifdef __cplusplus
}
#endif
#endif /* OMNI_AMD_H */
