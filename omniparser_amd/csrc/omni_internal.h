// Internal helpers shared by the gfx950 kernels of libomni_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/omni_amd.h"

void omni_set_error(const char* fmt, ...);

#define OMNI_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      omni_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return OMNI_E_HIP;                                                            \
    }                                                                               \
  } while (0)

#define OMNI_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      omni_set_error(__VA_ARGS__);              \
      return OMNI_E_ARG;                        \
    }                                           \
  } while (0)

typedef _Float16 half_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kVec = 4;   // elements per 16-byte vector
  static __host__ __device__ __forceinline__ float to_f32(float v) { return v; }
  static __host__ __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct ElemTraits<half_t> {
  static constexpr int kVec = 8;
  static __host__ __device__ __forceinline__ float to_f32(half_t v) { return (float)v; }
  static __host__ __device__ __forceinline__ half_t from_f32(float v) { return (half_t)v; }
};

// per-kind launchers (each lives in its own .hip file)
int omni_launch_conv(const omni_op_t* op, hipStream_t s);
int omni_launch_avgpool2(const omni_op_t* op, hipStream_t s);
int omni_launch_maxpool(const omni_op_t* op, hipStream_t s);
int omni_launch_resize_nearest(const omni_op_t* op, hipStream_t s);
int omni_launch_letterbox(const omni_op_t* op, hipStream_t s);
int omni_launch_detect_decode(const omni_op_t* op, hipStream_t s);
int omni_launch_nms(const omni_op_t* op, hipStream_t s);
int omni_launch_dwconv3(const omni_op_t* op, hipStream_t s);
int omni_launch_layernorm(const omni_op_t* op, hipStream_t s);
int omni_launch_dwconv3_ln(const omni_op_t* op, hipStream_t s);
int omni_launch_attention(const omni_op_t* op, hipStream_t s);
int omni_launch_misc(const omni_op_t* op, hipStream_t s);
