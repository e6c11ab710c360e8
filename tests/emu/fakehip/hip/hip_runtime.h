// TEST INFRASTRUCTURE — a host emulation of the slice of HIP / gfx950 that omniparser_amd/csrc/*.hip uses, so that the DEVICE
// SOURCES of the kernels can be compiled with a host C++ compiler and executed on a box without a GPU (tests/emu/build_emu.py,
// tests/test_kernels_emu_cpu.py).  It is found instead of the real <hip/hip_runtime.h> because tests/emu/fakehip is first on the
// include path.  Nothing in the product includes or links it.
//
// Execution model: every work-item is a FIBER (its own stack, a dozen-instruction context switch); all fibers of one workgroup run on
// one OS thread, round-robin, switching only at synchronisation points; workgroups are spread over the host cores.  `__shared__`
// is function-local `static thread_local` storage (= per workgroup, since a workgroup lives on one thread).  Cross-lane operations
// (MFMA, __shfl*, readlane) are collectives over a wave (64 consecutive work-items): publish, wait until every live lane of the
// wave has published, read.  __syncthreads / s_barrier waits for every live work-item of the workgroup.  LDS-DMA
// (buffer_load ... lds) completes at the issuing work-item's next sufficient `s_waitcnt vmcnt`, so a missing wait reads stale LDS
// here as it may on the hardware.  A work-item that returns leaves the wave and the workgroup.
// MFMA operand / result maps: 32x32 results col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); 16x16 results
// col = lane & 15, row = 4 (lane >> 4) + r; A/B fragments hold 8 consecutive k for row/col lane & 31 (resp. & 15), k block
// lane >> 5 (resp. >> 4) — the maps the MI355X-validated kernels of this repo were written against: if they were wrong here, those
// kernels would fail under emulation.
#pragma once
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <sys/mman.h>
#include <thread>
#include <type_traits>
#include <vector>

#define __HIP_DEVICE_COMPILE__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define address_space(n)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };

namespace emu {

// ---- fibers: callee-saved registers + stack pointer (x86-64 SysV); emu_switch(save, load) parks the caller and resumes `load`
extern "C" void emu_switch(void** save_sp, void* load_sp);
__asm__(R"(
.pushsection .text
.weak emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
.popsection
)");

constexpr int SLOT = 128;                     // bytes a lane can publish in one collective
constexpr size_t STACK = 192 * 1024;          // per work-item
struct Dma { unsigned char* dst; unsigned char data[16]; int bytes; };
struct Fiber {
  void* sp = nullptr;
  dim3 tid;
  int lane = 0, wave = 0, idx = 0;
  bool alive = false;
  unsigned coll = 0;                          // collectives this lane has entered
  std::deque<Dma> dma;
};
struct Wave {
  alignas(64) unsigned char buf[2][64][SLOT];
  unsigned tag[2];
  int count[2];
  int alive;
};
struct Block {
  dim3 bid, bdim, gdim;
  int n = 0, alive = 0, arrived = 0;
  unsigned gen = 0;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  unsigned char* dyn_lds = nullptr;
  void* sched_sp = nullptr;
  void (*body)(void*) = nullptr;
  void* body_arg = nullptr;
};
inline thread_local Block* blk = nullptr;
inline thread_local Fiber* cur = nullptr;

// OMNI_EMU_ORDER=reverse runs the work-items of a workgroup in descending order between synchronisation points (default: ascending):
// a missing barrier between "write my slot" and "read a neighbour's slot" shows under at least one of the two orders
inline bool reverse_order() {
  static const bool r = [] { const char* e = std::getenv("OMNI_EMU_ORDER"); return e && e[0] == 'r'; }();
  return r;
}
inline int next_index(int i, int n) { return reverse_order() ? (i == 0 ? n - 1 : i - 1) : (i + 1 == n ? 0 : i + 1); }

inline void yield() {
  Block& b = *blk;
  int nxt = cur->idx;
  do { nxt = next_index(nxt, b.n); } while (!b.fibers[nxt].alive);
  if (nxt == cur->idx) return;
  Fiber* from = cur;
  cur = &b.fibers[nxt];
  emu_switch(&from->sp, cur->sp);
}

inline void wait_vmcnt(int n) {
  std::deque<Dma>& q = cur->dma;
  while ((int)q.size() > n) {
    Dma& d = q.front();
    std::memcpy(d.dst, d.data, d.bytes);
    q.pop_front();
  }
}

inline void block_barrier() {
  Block& b = *blk;
  if (++b.arrived >= b.alive) { b.arrived = 0; ++b.gen; return; }
  const unsigned g = b.gen;
  while (b.gen == g) yield();
}

// publish `v` (<= SLOT bytes), let `f(lane -> T)` read any lane's value once every live lane of the wave has published
template <class T, class F>
inline auto collective(const T& v, F&& f) {
  static_assert(sizeof(T) <= SLOT, "wave scratch");
  Fiber* me = cur;
  Wave& w = blk->waves[me->wave];
  const unsigned seq = me->coll++;
  const int q = seq & 1;
  if (w.tag[q] != seq) { w.tag[q] = seq; w.count[q] = 0; }       // first lane of this collective (every lane is done reading seq - 2)
  std::memcpy(w.buf[q][me->lane], &v, sizeof(T));
  ++w.count[q];
  while (w.tag[q] == seq && w.count[q] < w.alive) yield();
  return f([&w, q](int l) { T t; std::memcpy(&t, w.buf[q][l & 63], sizeof(T)); return t; });
}

// same, but `f` gets lane -> const T* into the wave's buffer (valid until this lane's next collective): no copies
template <class T, class F>
inline auto collective_ptr(const T& v, F&& f) {
  static_assert(sizeof(T) <= SLOT && alignof(T) <= 64, "wave scratch");
  Fiber* me = cur;
  Wave& w = blk->waves[me->wave];
  const unsigned seq = me->coll++;
  const int q = seq & 1;
  if (w.tag[q] != seq) { w.tag[q] = seq; w.count[q] = 0; }
  std::memcpy(w.buf[q][me->lane], &v, sizeof(T));
  ++w.count[q];
  while (w.tag[q] == seq && w.count[q] < w.alive) yield();
  return f([&w, q](int l) { return reinterpret_cast<const T*>(w.buf[q][l & 63]); });
}

[[noreturn]] inline void fiber_exit() {
  Block& b = *blk;
  Fiber* me = cur;
  wait_vmcnt(0);
  me->alive = false;
  --b.waves[me->wave].alive;
  --b.alive;
  if (b.alive > 0 && b.arrived >= b.alive) { b.arrived = 0; ++b.gen; }     // the others were waiting for this work-item only
  if (b.alive == 0) {
    void* dummy;
    emu_switch(&dummy, b.sched_sp);
  } else {
    int nxt = me->idx;
    do { nxt = next_index(nxt, b.n); } while (!b.fibers[nxt].alive);
    cur = &b.fibers[nxt];
    void* dummy;
    emu_switch(&dummy, cur->sp);
  }
  std::abort();
}

inline void fiber_entry() {
  Block& b = *blk;
  b.body(b.body_arg);
  fiber_exit();
}

struct Worker {                               // one host thread: stacks for the largest workgroup, reused block after block
  unsigned char* stacks = nullptr;
  size_t n_stacks = 0;
  std::vector<unsigned char> dyn;
  Block b;
  ~Worker() { if (stacks) munmap(stacks, n_stacks * STACK); }
  void run_block(dim3 bid, dim3 bdim, dim3 gdim, size_t shmem, void (*body)(void*), void* arg) {
    const int n = (int)(bdim.x * bdim.y * bdim.z);
    if ((size_t)n > n_stacks) {
      if (stacks) munmap(stacks, n_stacks * STACK);
      n_stacks = n;
      stacks = (unsigned char*)mmap(nullptr, n_stacks * STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (stacks == MAP_FAILED) { std::perror("emu: mmap"); std::abort(); }
    }
    if (dyn.size() < shmem + 64) dyn.resize(shmem + 64);
    b.bid = bid; b.bdim = bdim; b.gdim = gdim;
    b.n = b.alive = n; b.arrived = 0; b.gen = 0;
    b.dyn_lds = dyn.data() + (64 - reinterpret_cast<uintptr_t>(dyn.data()) % 64) % 64;
    b.body = body; b.body_arg = arg;
    b.fibers.resize(n);
    b.waves.resize((n + 63) / 64);
    for (int w = 0; w * 64 < n; ++w) {
      b.waves[w].tag[0] = b.waves[w].tag[1] = ~0u;
      b.waves[w].count[0] = b.waves[w].count[1] = 0;
      b.waves[w].alive = std::min(64, n - w * 64);
    }
    for (int t = 0; t < n; ++t) {
      Fiber& f = b.fibers[t];
      f.tid = dim3(t % bdim.x, (t / bdim.x) % bdim.y, t / (bdim.x * bdim.y));
      f.lane = t & 63; f.wave = t >> 6; f.idx = t; f.alive = true; f.coll = 0;
      f.dma.clear();
      void** top = reinterpret_cast<void**>(stacks + (size_t)(t + 1) * STACK);     // 16-byte aligned
      top[-1] = nullptr;                                  // where fiber_entry's caller's frame would be
      top[-2] = reinterpret_cast<void*>(&fiber_entry);    // `ret` target: rsp % 16 == 8 on entry, as after a call
      for (int k = 3; k <= 8; ++k) top[-k] = nullptr;     // rbp rbx r12 r13 r14 r15
      f.sp = top - 8;
    }
    blk = &b;
    cur = &b.fibers[reverse_order() ? n - 1 : 0];
    emu_switch(&b.sched_sp, cur->sp);         // returns when the last work-item has exited
    blk = nullptr; cur = nullptr;
  }
};

inline int host_threads() {
  static const int n = [] {
    const char* e = std::getenv("OMNI_EMU_THREADS");
    int v = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
    return v > 0 ? v : 1;
  }();
  return n;
}

template <class F>
void launch(dim3 grid, dim3 block, size_t shmem, F&& body) {
  const long total = (long)grid.x * grid.y * grid.z;
  if (total <= 0) return;
  using Fn = std::remove_reference_t<F>;
  auto tramp = +[](void* p) { (*static_cast<Fn*>(p))(); };
  std::atomic<long> next{0};
  auto work = [&] {
    static thread_local Worker w;
    for (long i = next.fetch_add(1); i < total; i = next.fetch_add(1)) {
      const dim3 bid((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((long)grid.x * grid.y)));
      w.run_block(bid, block, grid, shmem, tramp, (void*)&body);
    }
  };
  const int nt = (int)std::min<long>(total, host_threads());
  if (nt <= 1) {
    std::thread(work).join();                 // never on the caller's (Python's) own stack / thread-locals
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back(work);
  for (auto& x : th) x.join();
}

}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::blk->bid)
#define blockDim (emu::blk->bdim)
#define gridDim (emu::blk->gdim)
#define __syncthreads() emu::block_barrier()
#define __builtin_amdgcn_s_barrier() emu::block_barrier()
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define OMNI_WAIT_VMCNT(n) emu::wait_vmcnt(n)
#define OMNI_WAIT_LGKM0() ((void)0)                  // LDS reads are synchronous here
#define OMNI_WAVE_SYNC() ((void)__shfl(0, 0))        // work-items are fibers: lanes of a wave need a real rendezvous
#define OMNI_DYN_LDS(type, name) type* name = reinterpret_cast<type*>(emu::blk->dyn_lds)

template <class A, class B> inline auto min(A a, B b) { using T = std::common_type_t<A, B>; return (T)a < (T)b ? (T)a : (T)b; }
template <class A, class B> inline auto max(A a, B b) { using T = std::common_type_t<A, B>; return (T)a > (T)b ? (T)a : (T)b; }
inline float __expf(float x) { return expf(x); }
#define __builtin_amdgcn_exp2f(x) exp2f(x)
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }

// agent-scope publish / ticket / read of the in-launch split-K combine (csrc/conv_igemm.hip): blocks are concurrent host threads,
// so the ticket carries the ordering the hardware recipe gets from write-through stores + vmcnt(0) + barrier (release on the way
// in, acquire on the way out); the data accesses are relaxed atomics like on the device
inline void emu_agent_st(float* p, float v) { __atomic_store(p, &v, __ATOMIC_RELAXED); }
inline float emu_agent_ld(const float* p) { float v; __atomic_load(const_cast<float*>(p), &v, __ATOMIC_RELAXED); return v; }
#define OMNI_AGENT_ST_F32(p, v) emu_agent_st((p), (v))
#define OMNI_AGENT_LD_F32(p) emu_agent_ld((p))
#define OMNI_AGENT_ADD_I32(p, v) __atomic_fetch_add((p), (v), __ATOMIC_ACQ_REL)
#define OMNI_AGENT_ST_I32(p, v) __atomic_store_n((p), (v), __ATOMIC_RELAXED)

inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }       // blocks run on concurrent host threads: a real fence
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) {
  unsigned* u = reinterpret_cast<unsigned*>(p);
  unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED), neu;
  float f;
  do { std::memcpy(&f, &old, 4); f += v; std::memcpy(&neu, &f, 4); } while (!__atomic_compare_exchange_n(u, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  std::memcpy(&f, &old, 4);
  return f;
}

// ---- cross-lane
template <class T> inline T __shfl(T v, int src, int width = 64) {
  const int base = emu::cur->lane & ~(width - 1);
  return emu::collective(v, [&](auto get) { return get(base + (src & (width - 1))); });
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
  const int l = emu::cur->lane, base = l & ~(width - 1);
  return emu::collective(v, [&](auto get) { return get(base + ((l ^ mask) & (width - 1))); });
}
inline int __any(int pred) {
  return emu::collective(pred, [&](auto get) { int r = 0; for (int q = 0; q < 64; ++q) r |= get(q) != 0; return r; });
}
template <class T> inline T emu_readlane(T v, int lane) { return emu::collective(v, [&](auto get) { return get(lane); }); }
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) emu_readlane((v), 0)

// ---- conversions
typedef __fp16 emu_hv2 __attribute__((ext_vector_type(2)));
inline unsigned short emu_f32_to_f16_rtz_bits(float x) {
  _Float16 h = (_Float16)x;                   // round to nearest even
  unsigned short b;
  std::memcpy(&b, &h, 2);
  const float back = (float)h;
  if (std::isinf(back) && !std::isinf(x)) return x < 0 ? 0xfbff : 0x7bff;
  if ((x > 0 && back > x) || (x < 0 && back < x)) b -= 1;      // rounded away from zero: one ulp back toward zero
  return b;
}
inline emu_hv2 emu_cvt_pkrtz(float a, float b) {
  const unsigned short ha = emu_f32_to_f16_rtz_bits(a), hb = emu_f32_to_f16_rtz_bits(b);
  const unsigned u = (unsigned)ha | ((unsigned)hb << 16);
  emu_hv2 r;
  std::memcpy(&r, &u, 4);
  return r;
}
#define __builtin_amdgcn_cvt_pkrtz(a, b) emu_cvt_pkrtz((a), (b))

// ---- MFMA
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
struct alignas(64) emu_ABf { float a[8], b[8]; };         // a lane's A / B fragment, widened once
inline emu_ABf emu_widen(emu_f16x8 a, emu_f16x8 b) {
  emu_ABf m;
  for (int i = 0; i < 8; ++i) { m.a[i] = (float)a[i]; m.b[i] = (float)b[i]; }
  return m;
}
inline emu_f32x16 emu_mfma_32x32x16_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x16 c, int, int, int) {
  const int l = emu::cur->lane;
  return emu::collective_ptr(emu_widen(a, b), [&](auto at) {
    emu_f32x16 d = c;
    const int col = l & 31;
    const emu_ABf *b0 = at(col), *b1 = at(col + 32);
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      const emu_ABf *a0 = at(row), *a1 = at(row + 32);
      double s = c[r];                          // products exact, one rounding per instruction (the hardware sums a k block wide)
      for (int k = 0; k < 8; ++k) s += (double)(a0->a[k] * b0->b[k]);
      for (int k = 0; k < 8; ++k) s += (double)(a1->a[k] * b1->b[k]);
      d[r] = (float)s;
    }
    return d;
  });
}
inline emu_f32x4 emu_mfma_16x16x32_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x4 c, int, int, int) {
  const int l = emu::cur->lane;
  return emu::collective_ptr(emu_widen(a, b), [&](auto at) {
    emu_f32x4 d = c;
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * (l >> 4) + r;
      double s = c[r];
      for (int kb = 0; kb < 4; ++kb) {
        const emu_ABf *ar = at(row + 16 * kb), *bc = at(col + 16 * kb);
        for (int k = 0; k < 8; ++k) s += (double)(ar->a[k] * bc->b[k]);
      }
      d[r] = (float)s;
    }
    return d;
  });
}
struct emu_AB1 { float a, b; };
inline emu_f32x16 emu_mfma_32x32x2_f32(float a, float b, emu_f32x16 c, int, int, int) {
  const int l = emu::cur->lane;
  return emu::collective(emu_AB1{a, b}, [&](auto get) {
    emu_AB1 fr[64];
    for (int q = 0; q < 64; ++q) fr[q] = get(q);
    emu_f32x16 d = c;
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      float s = c[r];
      for (int k = 0; k < 2; ++k) s += fr[row + 32 * k].a * fr[col + 32 * k].b;
      d[r] = s;
    }
    return d;
  });
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 emu_mfma_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 emu_mfma_16x16x32_f16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_32x32x2_f32

// ---- buffer resource + LDS-DMA
struct emu_rsrc { const unsigned char* base; };
typedef emu_rsrc __amdgpu_buffer_rsrc_t;
inline emu_rsrc emu_make_rsrc(void* p, int, int, int) { return emu_rsrc{(const unsigned char*)p}; }
#define __builtin_amdgcn_make_buffer_rsrc(p, s, n, f) emu_make_rsrc((void*)(p), (s), (n), (f))
inline void emu_buffer_load_lds(emu_rsrc r, void* lds, int bytes, unsigned voff, int soff, int ioff, int) {
  emu::Dma d;
  d.dst = (unsigned char*)lds + emu::cur->lane * bytes;            // M0 base + lane * size
  d.bytes = bytes;
  std::memcpy(d.data, r.base + voff + soff + ioff, bytes);
  emu::cur->dma.push_back(d);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, l, b, v, s, i, a) emu_buffer_load_lds((r), (void*)(l), (b), (v), (s), (i), (a))

// ---- runtime API (what csrc/*.hip calls)
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum { hipSuccess = 0, hipErrorNotSupported = 801 };
enum { hipStreamCaptureModeThreadLocal = 1 };
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "not supported by the host emulation"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 0; return hipSuccess; }       // no device: capi.hip's pointer check stays off
typedef void* hipDeviceptr_t;
inline hipError_t hipMemGetAddressRange(hipDeviceptr_t* base, size_t* size, hipDeviceptr_t) { *base = nullptr; *size = 0; return hipErrorNotSupported; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
// model_api.hip: device memory is host memory here
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1 };
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorNotSupported; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s_, size_t n, hipMemcpyKind) { std::memcpy(d, s_, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s_, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s_, n); return hipSuccess; }
// device globals are plain statics here (the range-guard counter of omni_internal.h)
#define HIP_SYMBOL(X) X
template <typename T> inline hipError_t hipMemcpyFromSymbol(void* d, const T& sym, size_t n, size_t off, hipMemcpyKind) {
  std::memcpy(d, reinterpret_cast<const char*>(&sym) + off, n); return hipSuccess;
}
template <typename T> inline hipError_t hipMemcpyToSymbol(T& sym, const void* s_, size_t n, size_t off, hipMemcpyKind) {
  std::memcpy(reinterpret_cast<char*>(&sym) + off, s_, n); return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s_, unsigned) { *s_ = nullptr; return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s_, uint32_t, const uint32_t*) { *s_ = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDebugDotPrint(hipGraph_t, const char*, unsigned) { return hipErrorNotSupported; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu::launch((grid), (block), (size_t)(shmem), [&] { kernel(__VA_ARGS__); })
