// Host emulation of omniparser_amd/csrc/glue_ops.hip::glue_kernel — TEST INFRASTRUCTURE (tests/test_glue_emu_cpu.py).
// The DEVICE SOURCE itself is compiled for the host: one workgroup = 256 std::threads, __syncthreads() = std::barrier,
// __shared__ = function-local statics (one workgroup runs at a time), atomicAdd = __atomic_fetch_add.  What this checks is the
// kernel's algorithm and its barrier placement (a missing __syncthreads() shows up as a data race here as it would on the GPU,
// if less reliably); what it cannot check is anything specific to the hardware (LDS capacity, launch, graphs).
#include <algorithm>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "../../include/omni_amd.h"

#define OMNI_HOST_EMU 1
#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(n)

namespace emu {
struct Idx { unsigned x; };
static thread_local Idx thread_idx;
static std::barrier<>* block_barrier = nullptr;
}  // namespace emu
#define threadIdx (emu::thread_idx)
#define __syncthreads() emu::block_barrier->arrive_and_wait()

using std::max;
using std::min;
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

#include "../../omniparser_amd/csrc/glue_ops.hip"

extern "C" const char* omni_emu_glue(const omni_op_t* op) {
  GlueArgs a;
  const char* why = nullptr;
  if (glue_args_from_op(op, a, &why)) return why;
  std::barrier<> bar(256);
  emu::block_barrier = &bar;
  std::vector<std::thread> th;
  th.reserve(256);
  for (unsigned t = 0; t < 256; ++t)
    th.emplace_back([t, a] {
      emu::thread_idx.x = t;
      glue_kernel(a);
    });
  for (auto& x : th) x.join();
  emu::block_barrier = nullptr;
  return nullptr;
}
