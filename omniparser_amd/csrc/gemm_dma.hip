// Pre-split LDS-DMA GEMM for gfx950: the captioner's linear layers (hf:models/florence2/modeling_florence2.py
// nn.Linear in DaViT blocks / projector, hf:models/bart/modeling_bart.py:143-308 attention + FFN projections).
//
//   Y[m, n] = act( 2^-k * sum_c A[m, c] * W'[n, c] + bias[n] ) (+ residual[m, n])         W' = W * 2^k
//
// Both operands arrive PRE-SPLIT in HBM as f16 pairs (4 bytes per element, the f32 footprint):
//   row = [K/16 groups][16 hi halves | 16 lo halves],   x = hi + lo,   hi = f16(x) (rtz),  lo = f16(x - hi)
// ("format B": lo is NOT rescaled, so ONE f32 accumulator takes all three products hi*hi + hi*lo + lo*hi; the weights
// are pre-multiplied by a power of two so that their lo parts stay normal f16 numbers, activations are O(1) and keep
// ~22 bits down to |x| = 2^-3 and an absolute 2^-25 floor below).  Producers (LayerNorm, the previous GEMM's epilogue,
// split_convert_kernel) write that format, so the K loop holds no conversion, no VGPR staging and no ds_write:
//   * global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB = 8 rows x 128 B per wave instruction), SRD + SGPR
//     K offset + one VGPR row offset per instruction: zero VALU address work inside the loop;
//   * LDS rows are unpadded 128 B (one 32-wide K slice); bank conflicts are avoided by an XOR swizzle of the 16-byte
//     chunk index with (row >> 1) & 7, applied to the DMA SOURCE address and to the fragment READ address (same
//     involution; symbolic check in tests/test_split_kernel_layout_cpu.py);
//   * NSTAGE-deep LDS ring, one raw s_barrier per K slice, counted s_waitcnt vmcnt so later slices stay in flight;
//   * MFMA fragments are double-buffered in registers (next unit's ds_read_b128 issue under the current 12 MFMAs);
//   * D^T = W * A^T: the MFMA's row operand is the weight fragment, so every lane owns ONE token and 4 consecutive
//     output channels per accumulator quad -> 16-byte residual loads / stores, split output as two 8-byte stores;
//   * epilogue is compiled per (activation, output format, residual) — no per-element branches.
// Ablations that led here (profiles/r2_gemm_diag.md): the round-1 kernel was bound by its compute side
// (ds_read -> lgkmcnt(0) -> MFMA without prefetch, 10.8 VALU per MFMA, a per-element epilogue), not by memory.
#include "omni_internal.h"
#include "gemm_common.h"
#include <string.h>
#include <type_traits>

OMNI_RANGE_GUARD_TU()

namespace {

typedef __attribute__((address_space(3))) void lds_void;

struct GemmArgs {
  const unsigned char* x; const unsigned char* w; const float* bias; const float* res; unsigned char* y;
  int M, N, K, nk;
  int ldi, in_coff, ldo, out_coff, ldr, res_coff;
  int mtiles, ntiles, xcd_order, xcd_n;
  float oscale;                       // 2^-k of the weight pre-scale
};

// Epilogue shared by the kernels below.
// D^T layout: lane -> token (lane & 31) of each token tile; accumulator quad q of channel tile j holds
// channels j*32 + 8q + 4*(lane >> 5) + 0..3.  A lane therefore owns 16-byte pieces of 32 different rows: stored directly, one
// store instruction scatters 32 x 32 bytes (the write-heavy layers — fc1, qkv of DaViT stages 0-1 — ran at 2.7-2.9 TB/s while
// read-heavy ones streamed at 5.1, profiles/r3_s2b_caption_per_op.txt).  So each wave transposes one 32-row token tile at a time
// through its own slice of the (now idle) LDS ring: rows are assembled in LDS exactly as they lie in memory (f32, or format B as
// two 8-byte halves), then 16 lanes read one row's 256 bytes and store them contiguously — 4 rows x 256 B per instruction; the
// residual is added after the transposition, from equally coalesced loads.
template <int BM, int BN, int WM, int WN, int TM, int TN, int LDS_BYTES, int ACT, bool OSPLIT, bool RES>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TM][TN], const GemmArgs& a, unsigned char* lds, int m0, int n0, int wave, int lane) {
  constexpr int NW = WM * WN;
  const int wm = wave / WN, wn = wave % WN;
  const int hsel = lane >> 5;
  static_assert(TN % 2 == 0, "epilogue stages 64 channels (256 bytes per row) of a wave at a time");
  constexpr int EP = 272;                                    // staged row pitch: 256 + 16 (conflict-free 16-byte column writes)
  static_assert(NW * 32 * EP <= LDS_BYTES, "epilogue staging must fit the LDS ring");
  __syncthreads();                                           // every wave is done reading the last K slice
  unsigned char* stg = lds + wave * (32 * EP);
  const float osc = a.oscale;
  float amax = 0.0f;                                         // range guard of the split output (omni_internal.h)
  const int rrow = lane >> 4, chunk = lane & 15;             // read phase: 16 lanes per row, 16 bytes per lane
  static_assert(!(RES && OSPLIT), "residual + format-B output is not a layer of this model");
#pragma unroll
  for (int jh = 0; jh < TN / 2; ++jh) {
    const int nw0 = n0 + wn * (BN / WN) + jh * 64;           // first channel of this 64-channel pass
    f32x4 bq[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        bq[j][q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nw0 + 4 * hsel + j * 32 + q * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 rres[8];
    auto load_res = [&](int i) {                             // residual rows of token tile i in the read-phase layout (rows clamped: M tail)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = min(m0 + wm * (BM / WM) + i * 32 + it * 4 + rrow, a.M - 1);
        rres[it] = *reinterpret_cast<const f32x4*>(a.res + (long long)m * a.ldr + a.res_coff + nw0 + chunk * 4);
      }
    };
    if constexpr (RES) load_res(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      unsigned char* wr = stg + (lane & 31) * EP;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x16& ac = acc[i][jh * 2 + j];
          float v[4];
          if constexpr (ACT == OMNI_ACT_GELU) {              // two elements per packed instruction
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
              const f32x2 g = omni_gelu2(f32x2{ac[q * 4 + c], ac[q * 4 + c + 1]} * osc + f32x2{bq[j][q][c], bq[j][q][c + 1]});
              v[c] = g[0]; v[c + 1] = g[1];
            }
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float t = ac[q * 4 + c] * osc + bq[j][q][c];
              if constexpr (ACT == OMNI_ACT_SILU) t = t / (1.0f + expf(-t));
              v[c] = t;
            }
          }
          const int cl = j * 32 + q * 8 + 4 * hsel;          // channel within the pass's 64
          if constexpr (OSPLIT) {
            uint2 hi, lo;
            omni_split4(v, hi, lo, amax);
            *reinterpret_cast<uint2*>(wr + omni_split_off(cl)) = hi;
            *reinterpret_cast<uint2*>(wr + omni_split_off(cl) + 32) = lo;
          } else {
            *reinterpret_cast<f32x4*>(wr + cl * 4) = f32x4{v[0], v[1], v[2], v[3]};
          }
        }
      OMNI_WAVE_SYNC();
      const int mt0 = m0 + wm * (BM / WM) + i * 32;
      f32x4 ov[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        ov[it] = *reinterpret_cast<const f32x4*>(stg + (it * 4 + rrow) * EP + chunk * 16);
        if constexpr (RES) ov[it] += rres[it];
      }
      if constexpr (RES) { if (i + 1 < TM) load_res(i + 1); }  // next tile's residual rows: in flight under the stores and the next tile's arithmetic
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = mt0 + it * 4 + rrow;
        if (m < a.M) *reinterpret_cast<f32x4*>(a.y + ((long long)m * a.ldo + a.out_coff + nw0) * 4 + chunk * 16) = ov[it];
      }
      OMNI_WAVE_SYNC();                                      // the next token tile overwrites the staging rows
    }
  }
  if constexpr (OSPLIT) omni_report_range(amax);
}

// K-loop schedule: the DMA pieces of the slice NSTAGE-1 ahead are spread over the units, ds_reads / DMA pieces interleaved one per
// MFMA (sched_group_barrier).  The alternative schedule (all DMA pieces right after the barrier: 1-6 % slower) and the ablation
// variants that led here are recorded in profiles/r2_gemm_diag.md; they are not compiled into the library.
// SCHED 1 (the 256x256 tile; SCHED 0 = pieces spread over all units, the smaller tiles): the DMA pieces of the next slice are issued
// in the FIRST half of the units, so that the last piece has at least half a slice to land before the wait at the top of the next
// slice: +2-5 % on the K >= 512 shapes (profiles/r3_s3_gemm_bench.txt).  Measured and NOT kept (within +-5 % of this kernel, removed
// from the library after commit f6047af; profiles/r3_s7_gemm_bench_sched2_w4.txt, r3_s8_gemm_bench_k16.txt): the barrier in front of
// the last unit of a slice, a 4-wave 256x256 tile, a 4-wave 256x128 tile with 16-wide K slices and two blocks per CU.
template <int BM, int BN, int WM, int WN, int NSTAGE, int ACT, bool OSPLIT, bool RES, int SCHED = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_dma_kernel(GemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type / LDS-DMA builtins do not exist in the host pass (it only needs the stub)
  constexpr int NW = WM * WN;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);     // 32x32 MFMA tiles per wave (tokens x channels)
  constexpr int A_DMA = BM / (8 * NW), B_DMA = BN / (8 * NW);  // 8-row DMA pieces per wave and K slice
  constexpr int DPS = A_DMA + B_DMA;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NP = TM / 2;                                   // token-tile pairs per wave
  constexpr int NU = 2 * NP;                                   // MFMA units per K slice: (16-wide K group, tile pair)
  constexpr int NM = 6 * TN;                                   // MFMAs per unit: 2 token tiles x TN channel tiles x 3 products
  static_assert(TM % 2 == 0 && TN >= 1 && A_DMA >= 1 && B_DMA >= 1 && NSTAGE >= 2 && NSTAGE <= 4, "tile / wave grid mismatch");
  static_assert(BM * 128 + (TN - 1) * 4096 < 65536 && (TM - 1) * 4096 < 65536, "ds_read immediate offsets");
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // Measured and not kept (round 4, profiles/r4_s3_persistent_gemm_ab.txt): persistent blocks walking the tile grid with stride G, to cap
  // the CUs a GEMM occupies and leave the rest to the other lane's HBM-bound kernels.  Bit-identical results; G = 256: 665 vs 672 ms
  // per bench step (noise), G = 224 / 192 / 160 / 128: 669 / 687 / 700 / 725 — the GEMM alone slows by 5 / 11 / 22 / 41 % and the
  // overlap wins back only part of it.  The loop also cost ~20 VGPRs.
  int mt, nt;
  if (!tile_of_block(blockIdx.x, a.mtiles, a.ntiles, a.xcd_order, a.xcd_n, mt, nt)) return;
  // block-uniform by construction; said explicitly so that the buffer resources below are built in SGPRs (the division inside
  // tile_of_block runs on the VALU, and a resource the compiler believes divergent costs a waterfall loop per LDS-DMA instruction)
  const int m0 = __builtin_amdgcn_readfirstlane(mt) * BM, n0 = __builtin_amdgcn_readfirstlane(nt) * BN;

  // ---- LDS-DMA descriptors.  Piece p (8 rows x 128 B) of an operand tile lands at LDS piece slot p; lane l of the
  // issuing wave supplies row 8p + l/8, 16-byte slot l%8, whose SOURCE chunk is slot ^ ((row >> 1) & 7).
  const int rsub = lane >> 3, slot = lane & 7;
  unsigned voffA[A_DMA], voffB[B_DMA];
  // which 8-row piece of the activation tile wave `wave` stages as its i-th: contiguous bands (SCHED 0 / 1), or — SCHED 2 — strided by
  // the wave count, so that every wave's pieces 0, 2 are rows of the FIRST 64-token half of a wave tile (wm * 128 + [0, 64)) and its
  // pieces 1, 3 rows of the second half: the ping-pong schedule stages the halves at different times (see `slice_pp` below)
  auto pieceA = [&](int i) { return SCHED == 2 ? wave + NW * i : wave * A_DMA + i; };
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    const int rl = pieceA(i) * 8 + rsub;
    const int rr = min(rl, a.M - 1 - m0);                  // M tail: re-read the last row (its results are never stored)
    voffA[i] = (unsigned)rr * (unsigned)(a.ldi * 4) + (unsigned)((slot ^ ((rl >> 1) & 7)) * 16);
  }
#pragma unroll
  for (int i = 0; i < B_DMA; ++i) {
    const int rl = (wave * B_DMA + i) * 8 + rsub;
    const int rr = min(rl, a.N - 1 - n0);
    voffB[i] = (unsigned)rr * (unsigned)(a.K * 4) + (unsigned)((slot ^ ((rl >> 1) & 7)) * 16);
  }
  const __amdgpu_buffer_rsrc_t rsrcA =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + ((long long)m0 * a.ldi + a.in_coff) * 4), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.w + (long long)n0 * a.K * 4), 0, 0x7fffffff, 0x00020000);
  // piece i of slice kt into ring slot `stage` (i < A_DMA: activation rows, else weight rows)
  auto issue_piece = [&](int kt, int stage, int i) {
    const int so = kt * 128;
    if (i < A_DMA)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(lds + stage * STAGE + pieceA(i) * 1024), 16, voffA[i], so, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(lds + stage * STAGE + BM * 128 + (wave * B_DMA + i - A_DMA) * 1024), 16,
                                               voffB[i - A_DMA], so, 0, 0);
  };
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < DPS; ++i) issue_piece(kt, stage, i);
  };

  // ---- fragment read offsets: lane -> (row lane & 31, k half lane >> 5); chunk of (K group g, part p) = 4g + 2p + half
  const int swz = (lane >> 1) & 7, hsel = lane >> 5;
  int offA[4], offW[4];
#pragma unroll
  for (int gp = 0; gp < 4; ++gp) {
    const int c = ((((gp >> 1) * 4 + (gp & 1) * 2 + hsel) ^ swz)) * 16;
    offA[gp] = (wm * (BM / WM) + (lane & 31)) * 128 + c;
    offW[gp] = BM * 128 + (wn * (BN / WN) + (lane & 31)) * 128 + c;
  }
  struct AF { f16x8 h[2], l[2]; };
  struct WF { f16x8 h[TN], l[TN]; };
  auto loadA = [&](const unsigned char* st, int g, int ip, AF& f) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f.h[t] = *reinterpret_cast<const f16x8*>(st + offA[g * 2 + 0] + (ip * 2 + t) * 4096);
      f.l[t] = *reinterpret_cast<const f16x8*>(st + offA[g * 2 + 1] + (ip * 2 + t) * 4096);
    }
  };
  auto loadW = [&](const unsigned char* st, int g, WF& f) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f.h[j] = *reinterpret_cast<const f16x8*>(st + offW[g * 2 + 0] + j * 4096);
      f.l[j] = *reinterpret_cast<const f16x8*>(st + offW[g * 2 + 1] + j * 4096);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // three products per accumulator, issued product-major so that the same accumulator recurs every 2*TN MFMAs.
  // What the third product buys (round-5 ablation on the MI355X, tools/archive/r5_gemm_products_ablation.{patch,py},
  // profiles/r5_s2_products_ablation.jsonl): with w_lo*x_hi or w_hi*x_lo dropped on every 256x256-tile layer the GEMM family of a
  // 128-crop encode costs 142.8 / 144.5 instead of 160.6 ms (-11 %: the kernel is power / bandwidth bound, a third fewer MFMAs is
  // not a third less time) and the greedy caption ids change on 53 / 61 of 735 crops; restricted to the K = 2048 layers: -2 %, 32 / 24
  // crops.  Token-exactness against the f32 CPU path needs all three; the knob was removed after the measurement.
  auto mma = [&](const AF& af, const WF& wf, int ip) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[ip * 2 + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h[j], af.h[t], acc[ip * 2 + t][j], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[ip * 2 + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.l[j], af.h[t], acc[ip * 2 + t][j], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[ip * 2 + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h[j], af.l[t], acc[ip * 2 + t][j], 0, 0, 0);
  };

  const int nk = a.nk;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s, s);
  int stage = 0, nstage = NSTAGE - 1;          // ring positions of slice kt and of slice kt + NSTAGE - 1
  constexpr int IU = SCHED == 1 ? (NU + 1) / 2 : NU;   // units that issue DMA pieces
  constexpr int PPU = (DPS + IU - 1) / IU;     // DMA pieces per issuing unit
  AF af[2];
  WF wf[2];
  // One K slice.  MORE (compile time): a further slice is issued into the ring — the steady-state loop and the drain loop are
  // separate copies so that no branch splits the scheduling region (ds_reads / DMA pieces interleave with the MFMAs).
  auto slice = [&](int kt, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    // slice kt landed (this wave's pieces), later slices may stay in flight; then everybody's pieces landed and everybody is
    // done reading the stage that the next DMA overwrites (it was read in iteration kt - 1)
    if constexpr (NSTAGE > 2 && MORE) OMNI_WAIT_VMCNT((NSTAGE - 2) * DPS);
    else OMNI_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    const unsigned char* st = lds + stage * STAGE;
    loadW(st, 0, wf[0]);
    loadA(st, 0, 0, af[0]);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int g = u / NP, ip = u % NP;
      int nds = 0;
      if (u + 1 < NU) {
        const int g2 = (u + 1) / NP, ip2 = (u + 1) % NP;
        if (g2 != g) { loadW(st, g2, wf[g2 & 1]); nds += 2 * TN; }
        loadA(st, g2, ip2, af[(u + 1) & 1]);
        nds += 4;
      }
      if constexpr (MORE) {
#pragma unroll
        for (int i = u * PPU; i < (u + 1) * PPU && i < DPS; ++i) issue_piece(kt + NSTAGE - 1, nstage, i);
      }
      mma(af[u & 1], wf[g & 1], ip);
      {
        // one ds_read (then one DMA piece) behind each of the first MFMAs: they issue inside the 32-cycle MFMA slots
        const int npc = (MORE && u < IU) ? PPU : 0;
#pragma unroll
        for (int k = 0; k < NM; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          if (k < nds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          else if (k < nds + npc) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = stage + 1 == NSTAGE ? 0 : stage + 1;
    nstage = nstage + 1 == NSTAGE ? 0 : nstage + 1;
  };
  // ---- SCHED 2 (round 6): PING-PONG.  The two waves of a SIMD (wave w and w + 4 = the two M halves of the tile) run HALF A PHASE
  // apart: a K slice is four (load segment | barrier | 12 MFMAs | barrier) phases, and while waves 0-3 issue their MFMAs waves 4-7
  // issue the fragment reads and LDS-DMA pieces of THEIR next phase — the matrix pipe of a SIMD is fed by one wave at a time, the
  // other wave's LDS / DMA issue never competes with MFMA issue (MI355X_MICROARCH "two waves per SIMD", the guide's 8-phase GEMM
  // template).  Units are token-half major: u0 = (K group 0, tokens 0-63), u1 = (g1, 0-63), u2 = (g0, 64-127), u3 = (g1, 64-127);
  // the weight fragments of both K groups stay in registers after u0 / u1.  Staging of slice kt + 1 during slice kt, per wave:
  // u0 its two FIRST-half activation pieces, u1 / u2 two weight pieces each, u3 its two SECOND-half activation pieces;
  // `vmcnt(2)` in u3 retires everything but those last two, which are retired by `vmcnt(4)` in u1 of the NEXT slice — two barriers
  // before u2 reads them (an LDS-DMA is ordered for another wave's ds_read only by the issuer's vmcnt followed by a barrier both
  // have passed).  Nothing is restaged earlier than four phases after its last read.  Same products in the same order per
  // accumulator as SCHED 0 / 1: bit-identical results.
  if constexpr (SCHED == 2) {
    static_assert(BM == 256 && BN == 256 && WM == 2 && WN == 4 && NSTAGE == 2 && TM == 4 && TN == 2 && A_DMA == 4 && B_DMA == 4, "ping-pong: 256x256 tile, 8 waves");
    // prologue of the generic path issued slice 0 (all eight pieces per wave): land it, publish it, then stagger the M halves
    OMNI_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    auto phase_barrier = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    {
      AF afp;
      auto slice_pp = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const unsigned char* st = lds + stage * STAGE;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int g = u & 1, ip = u >> 1;
          if (u < 2) loadW(st, g, wf[g]);
          loadA(st, g, ip, afp);
          if constexpr (MORE) {
            if (u == 0) { issue_piece(kt + 1, nstage, 0); issue_piece(kt + 1, nstage, 2); }
            if (u == 1) { issue_piece(kt + 1, nstage, A_DMA + 0); issue_piece(kt + 1, nstage, A_DMA + 1); }
            if (u == 2) { issue_piece(kt + 1, nstage, A_DMA + 2); issue_piece(kt + 1, nstage, A_DMA + 3); }
            if (u == 3) { issue_piece(kt + 1, nstage, 1); issue_piece(kt + 1, nstage, 3); }
            if (u == 1) OMNI_WAIT_VMCNT(4);          // this slice's second-half activation pieces (issued in u3 of the previous slice)
            if (u == 3) OMNI_WAIT_VMCNT(2);          // everything of the next slice but its second-half activation pieces
          } else {
            if (u == 1) OMNI_WAIT_VMCNT(0);
          }
          phase_barrier();
          __builtin_amdgcn_s_setprio(1);
          mma(afp, wf[g], ip);
          __builtin_amdgcn_s_setprio(0);
          phase_barrier();
        }
        stage ^= 1;
        nstage ^= 1;
      };
      int ktp = 0;
      for (; ktp + 1 < nk; ++ktp) slice_pp(ktp, std::true_type{});
      for (; ktp < nk; ++ktp) slice_pp(ktp, std::false_type{});
    }
    // Measured and not kept (round 6, profiles/r6_s5_gemm_sched.jsonl): the same ping-pong with TWO phases per slice (one per token half,
    // both K groups, 24 MFMAs between barriers) — 1.619 / 1.923 / 1.253 ms on fc2 / fc1 / qkv against 1.524 / 1.878 / 1.215 for the
    // four-phase form and 1.578 / 1.926 / 1.268 for the lockstep schedule: half the barriers, but sixteen fragment reads to wait for at
    // the head of every MFMA segment.
    if (wm == 0) __builtin_amdgcn_s_barrier();
  } else {
  // slices with a successor to issue: kt + NSTAGE - 1 < nk.  The NSTAGE > 2 counted wait needs the full ring in flight, which
  // holds exactly for those iterations.
  int kt = 0;
  for (; kt + NSTAGE - 1 < nk; ++kt) slice(kt, std::true_type{});
  for (; kt < nk; ++kt) slice(kt, std::false_type{});
  }

  // ---- epilogue (gemm_epilogue above): bias / activation / 2^-k in registers, rows transposed through the idle LDS ring
  gemm_epilogue<BM, BN, WM, WN, TM, TN, NSTAGE * STAGE, ACT, OSPLIT, RES>(acc, a, lds, m0, n0, wave, lane);
#endif
}

// f32 -> format B, elementwise over a channel slice of a token matrix (in place when x == y): used where a producer does
// not emit the split format itself (attention outputs, projector input).  One thread per 16-channel group.
__global__ __launch_bounds__(256) void split_convert_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, long long rows,
                                                            int C, int ldi, int in_coff, int ldo, int out_coff) {
  const int gpr = C / 16;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * gpr) return;
  const long long r = idx / gpr;
  const int g = (int)(idx - r * gpr);
  const float* src = x + r * ldi + in_coff + g * 16;
  f32x4 v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(src + q * 4);
  uint2 hi[4], lo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float t[4] = {v[q][0], v[q][1], v[q][2], v[q][3]};
    omni_split4(t, hi[q], lo[q]);
  }
  unsigned char* dst = y + (r * ldo + out_coff + g * 16) * 4;
  *reinterpret_cast<u32x4*>(dst) = u32x4{hi[0].x, hi[0].y, hi[1].x, hi[1].y};
  *reinterpret_cast<u32x4*>(dst + 16) = u32x4{hi[2].x, hi[2].y, hi[3].x, hi[3].y};
  *reinterpret_cast<u32x4*>(dst + 32) = u32x4{lo[0].x, lo[0].y, lo[1].x, lo[1].y};
  *reinterpret_cast<u32x4*>(dst + 48) = u32x4{lo[2].x, lo[2].y, lo[3].x, lo[3].y};
}

// ---- OMNI_OP_MLP_FUSED: the FFN of a DaViT block, y = x1 + fc2(GELU(fc1(h))), as ONE kernel for the narrow stage (C = 128, hidden
// 512: 4.7 M tokens per 128-crop micro-batch at 768x768).  As two GEMM launches its K = 128 / N = 128 layers are bound by everything
// except the matrix pipe: fc1 writes 9.7 GB of hidden activations (4 K slices of MFMA work per 256 KB tile written, GELU for 64 K
// elements with the pipe idle), fc2 reads them back — 154 and 231 TF/s where the K = 2048 layers reach 411
// (profiles/r3_s9_caption_per_op.txt).  Here a wave keeps its 32 tokens' input fragments (K = 128: 64 registers) and its 32 x 128
// output accumulators in registers and walks the hidden dimension in chunks of 32 channels:
//   acc1[32 hidden x 32 tokens]  = W1[chunk] . h^T              24 MFMAs (8 K groups x hi.hi + lo.hi + hi.lo)
//   g = GELU(acc1 * 2^-k1 + b1) -> (hi, lo) halves in registers
//   acc2[128 out x 32 tokens]   += W2[:, chunk] . g^T           24 MFMAs
// The D layout of the first product (lane = token, 4 consecutive hidden channels per accumulator quad) IS an MFMA column operand of
// the second one up to the order of the K index inside a 16-group: lane half h holds channels {4h..4h+3, 8+4h..8+4h+3}, so W2 is
// packed with its hidden axis permuted the same way inside every 16-group (planner.pack_weight_dma(kperm=True)) and the hidden
// activations never leave the register file.  Weight chunks (16 KB of W1 rows + 16 KB of W2 columns) stream L2 -> LDS by LDS-DMA
// into a two-stage ring shared by the block's four waves (same 128-byte rows, same XOR swizzle as gemm_dma_kernel); two blocks per
// CU, so one block's GELU / barrier overlaps the other's MFMAs.  HBM traffic: h in, residual in, y out — 12 bytes per element of
// the token matrix instead of 44.
struct MlpArgs {
  const unsigned char* x; const unsigned char* w1; const unsigned char* w2; const float* b1;
  GemmArgs ep{};                        // epilogue view: bias = b2, res, y, M, ldo, out_coff, ldr, res_coff, oscale = 2^-k2
  int ldi, in_coff;
  float osc1;                         // 2^-k1
};

template <int C, int HID>
__global__ __launch_bounds__(256, 2) void mlp_fused_kernel(MlpArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(C == 128 && HID % 32 == 0, "written for the C = 128 stage: 8 K groups in registers, 4 output tiles");
  constexpr int NW = 4, KG = C / 16, OB = C / 32, NCH = HID / 32;
  constexpr int W1B = 32 * C * 4, W2B = C * 128, STAGE = W1B + W2B;      // bytes of one chunk's W1 rows / W2 columns
  constexpr int P1 = W1B / 1024 / NW, P2 = W2B / 1024 / NW;              // 1 KiB DMA pieces per wave and chunk
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE + HID * 4];
  float* lb1 = reinterpret_cast<float*>(lds + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * (NW * 32);
  const int hsel = lane >> 5, swz = (lane >> 1) & 7;

  // ---- fc1 bias -> LDS (2 KB, read per chunk)
  for (int e = tid * 4; e < HID; e += 256 * 4) *reinterpret_cast<f32x4*>(lb1 + e) = *reinterpret_cast<const f32x4*>(a.b1 + e);

  // ---- LDS-DMA descriptors (piece = 8 rows x 128 B; lane l supplies row 8p + l/8, 16-byte slot l%8 from SOURCE chunk slot ^ swizzle)
  const int rsub = lane >> 3, slot = lane & 7;
  unsigned voff1[P1], voff2[P2];
#pragma unroll
  for (int i = 0; i < P1; ++i) {            // W1 chunk in LDS: [K slice s][32 hidden rows][128 B]; this wave: piece index wave * P1 + i
    const int pc = wave * P1 + i, rl = (pc & 3) * 8 + rsub, sl = pc >> 2;
    voff1[i] = (unsigned)rl * (unsigned)(C * 4) + (unsigned)(sl * 128) + (unsigned)((slot ^ ((rl >> 1) & 7)) * 16);
  }
#pragma unroll
  for (int i = 0; i < P2; ++i) {            // W2 chunk in LDS: [C output rows][128 B = this chunk's 32 hidden channels]
    const int rl = (wave * P2 + i) * 8 + rsub;
    voff2[i] = (unsigned)rl * (unsigned)(HID * 4) + (unsigned)((slot ^ ((rl >> 1) & 7)) * 16);
  }
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, 0x7fffffff, 0x00020000);
  auto issue = [&](int c, int stage) {
#pragma unroll
    for (int i = 0; i < P1; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (lds_void*)(lds + stage * STAGE + (wave * P1 + i) * 1024), 16, voff1[i], c * (32 * C * 4), 0, 0);
#pragma unroll
    for (int i = 0; i < P2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc2, (lds_void*)(lds + stage * STAGE + W1B + (wave * P2 + i) * 1024), 16, voff2[i], c * 128, 0, 0);
  };
  issue(0, 0);

  // ---- this wave's 32 tokens: all K = C input fragments stay in registers (lane -> token lane & 31, 8 halves 8 * (lane >> 5) .. of a group)
  f16x8 xh[KG], xl[KG];
  {
    const int m = min(m0 + wave * 32 + (lane & 31), a.ep.M - 1);             // M tail: re-read the last row (never stored)
    const unsigned char* xrow = a.x + ((long long)m * a.ldi + a.in_coff) * 4 + hsel * 16;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      xh[g] = *reinterpret_cast<const f16x8*>(xrow + g * 64);
      xl[g] = *reinterpret_cast<const f16x8*>(xrow + g * 64 + 32);
    }
  }
  // fragment read offsets inside a stage: row lane & 31, 16-byte chunk (group gg of the 32-wide slice, part hi / lo) = 4 gg + 2 part +
  // half, swizzled — four values; K slice (W1) and output tile (W2) are immediate offsets on top
  int fo[2][2];
#pragma unroll
  for (int gg = 0; gg < 2; ++gg)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) fo[gg][pt] = (lane & 31) * 128 + (((gg * 4 + pt * 2 + hsel) ^ swz) * 16);

  f32x16 acc2[1][OB];
#pragma unroll
  for (int j = 0; j < OB; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[0][j][e] = 0.0f;
  const float osc1 = a.osc1;
  float amax = 0.0f;                                     // range guard of the hidden activations' split (omni_internal.h)
  __syncthreads();                                       // fc1 bias visible

  for (int c = 0; c < NCH; ++c) {
    const int stage = c & 1;
    OMNI_WAIT_VMCNT(0);                                  // this wave's pieces of chunk c (and, the first time, its input fragments)
    __builtin_amdgcn_s_barrier();                        // everybody's pieces landed; everybody is done reading the other stage
    if (c + 1 < NCH) issue(c + 1, stage ^ 1);
    const unsigned char* st = lds + stage * STAGE;
    // ---- fc1 chunk: two accumulators (even / odd K groups) so that consecutive MFMAs do not wait for each other
    f32x16 a1[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { a1[0][e] = 0.0f; a1[1][e] = 0.0f; }
#pragma unroll
    for (int g = 0; g < KG; g += 2) {
      const unsigned char* sl = st + (g >> 1) * 4096;       // K slice g / 2 of the W1 rows; g even -> group 0, g + 1 -> group 1
      const f16x8 wh0 = *reinterpret_cast<const f16x8*>(sl + fo[0][0]), wl0 = *reinterpret_cast<const f16x8*>(sl + fo[0][1]);
      const f16x8 wh1 = *reinterpret_cast<const f16x8*>(sl + fo[1][0]), wl1 = *reinterpret_cast<const f16x8*>(sl + fo[1][1]);
      a1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xh[g], a1[0], 0, 0, 0);
      a1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xh[g + 1], a1[1], 0, 0, 0);
      a1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl0, xh[g], a1[0], 0, 0, 0);
      a1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl1, xh[g + 1], a1[1], 0, 0, 0);
      a1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xl[g], a1[0], 0, 0, 0);
      a1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xl[g + 1], a1[1], 0, 0, 0);
    }
    // ---- bias + GELU + split: accumulator e = 4q + r is hidden channel 32c + 8q + 4 * (lane >> 5) + r of token lane & 31
    f16x8 gh[2], gl[2];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      uint2 hq[2], lq[2];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = kg * 2 + qq;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(lb1 + c * 32 + 8 * q + 4 * hsel);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 t = omni_gelu2((f32x2{a1[0][q * 4 + r], a1[0][q * 4 + r + 1]} + f32x2{a1[1][q * 4 + r], a1[1][q * 4 + r + 1]}) * osc1 +
                                     f32x2{bq[r], bq[r + 1]});
          v[r] = t[0]; v[r + 1] = t[1];
        }
        omni_split4(v, hq[qq], lq[qq], amax);
      }
      gh[kg] = __builtin_bit_cast(f16x8, u32x4{hq[0].x, hq[0].y, hq[1].x, hq[1].y});
      gl[kg] = __builtin_bit_cast(f16x8, u32x4{lq[0].x, lq[0].y, lq[1].x, lq[1].y});
    }
    // ---- fc2 chunk: output tiles two at a time (their weight fragments are the live registers that decide the occupancy)
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int j = 0; j < OB; j += 2) {
        const unsigned char* sw = st + W1B + j * 4096;
        const f16x8 wh0 = *reinterpret_cast<const f16x8*>(sw + fo[kg][0]), wl0 = *reinterpret_cast<const f16x8*>(sw + fo[kg][1]);
        const f16x8 wh1 = *reinterpret_cast<const f16x8*>(sw + 4096 + fo[kg][0]), wl1 = *reinterpret_cast<const f16x8*>(sw + 4096 + fo[kg][1]);
        acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, gh[kg], acc2[0][j], 0, 0, 0);
        acc2[0][j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, gh[kg], acc2[0][j + 1], 0, 0, 0);
        acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl0, gh[kg], acc2[0][j], 0, 0, 0);
        acc2[0][j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl1, gh[kg], acc2[0][j + 1], 0, 0, 0);
        acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, gl[kg], acc2[0][j], 0, 0, 0);
        acc2[0][j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, gl[kg], acc2[0][j + 1], 0, 0, 0);
      }
  }
  omni_report_range(amax);
  // ---- epilogue: 2^-k2, fc2 bias, residual, coalesced f32 rows (gemm_epilogue: four waves x 32 tokens x 128 channels)
  gemm_epilogue<NW * 32, C, NW, 1, 1, OB, 2 * STAGE, OMNI_ACT_NONE, false, true>(acc2, a.ep, lds, m0, 0, wave, lane);
#endif
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
int launch_tile(GemmArgs& a, int act, int osplit, hipStream_t s) {
  a.mtiles = (a.M + BM - 1) / BM;
  a.ntiles = a.N / BN;
  a.xcd_order = (a.mtiles >= 64 && a.ntiles > 1) ? 1 : 0;
  a.xcd_n = a.xcd_order ? choose_xcd_n(a.ntiles, 4ll * a.N * a.K) : 1;
  dim3 grid(tile_grid(a.mtiles, a.ntiles, a.xcd_order, a.xcd_n)), block(WM * WN * 64);
  const bool res = a.res != nullptr;
  constexpr int SCHED1 = (BM == 256 && BN == 256) ? 1 : 0;     // front-loaded DMA issue on the 256x256 tile (see gemm_dma_kernel)
  // 256x256 tile: the ping-pong schedule (SCHED 2, round 6: -3.4 / -2.5 / -4.2 % on fc2 / fc1 / qkv of DaViT stage 2, bit-identical
  // results; 641 instead of 654-662 ms per bench step) unless OMNI_GEMM_SCHED=1 asks for the lockstep schedule (the A/B knob of
  // tools/gemm_exp.py sched and of tests/gpu_checks.py::check_gemm_schedules_bitwise)
  const char* sched_e = getenv("OMNI_GEMM_SCHED");
  const bool lockstep = sched_e && atoi(sched_e) == 1;
#define OMNI_GD(ACT_, OS_, RES_)                                                                                                   \
  do {                                                                                                                             \
    if constexpr (BM == 256 && BN == 256) {                                                                                        \
      if (!lockstep) { hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, NSTAGE, ACT_, OS_, RES_, 2>), grid, block, 0, s, a); break; }  \
    }                                                                                                                              \
    hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, NSTAGE, ACT_, OS_, RES_, SCHED1>), grid, block, 0, s, a);                  \
  } while (0)
  if (act == OMNI_ACT_NONE && !osplit && !res) OMNI_GD(OMNI_ACT_NONE, false, false);
  else if (act == OMNI_ACT_NONE && !osplit && res) OMNI_GD(OMNI_ACT_NONE, false, true);
  else if (act == OMNI_ACT_NONE && osplit && !res) OMNI_GD(OMNI_ACT_NONE, true, false);
  else if (act == OMNI_ACT_GELU && osplit && !res) OMNI_GD(OMNI_ACT_GELU, true, false);
  else if (act == OMNI_ACT_GELU && !osplit && !res) OMNI_GD(OMNI_ACT_GELU, false, false);
#undef OMNI_GD
  else {
    omni_set_error("gemm_dma: unsupported epilogue (act %d, split out %d, residual %d)", act, osplit, (int)res);
    return OMNI_E_ARG;
  }
  return OMNI_OK;
}

}  // namespace

// OMNI_OP_CONV with i20 == 2 (see include/omni_amd.h): pointwise, pre-split operands.
int omni_launch_gemm_dma(const omni_op_t* op, hipStream_t s) {
  GemmArgs a{};
  a.x = (const unsigned char*)op->p[0]; a.w = (const unsigned char*)op->p[1]; a.bias = (const float*)op->p[2];
  a.res = (const float*)op->p[3]; a.y = (unsigned char*)op->p[4];
  const int B = op->i[0], H = op->i[1], W = op->i[2];
  a.K = op->i[3]; a.ldi = op->i[4]; a.in_coff = op->i[5];
  a.N = op->i[12]; a.ldo = op->i[13]; a.out_coff = op->i[14];
  const int act = op->i[15];
  a.ldr = op->i[16]; a.res_coff = op->i[17];
  const int osplit = op->i[21];
  a.oscale = op->f[1] != 0.0f ? op->f[1] : 1.0f;
  OMNI_REQUIRE(op->dtype == OMNI_F32, "gemm_dma: f32 plans only");
  OMNI_REQUIRE(a.x && a.w && a.y, "gemm_dma: null pointer");
  OMNI_REQUIRE(op->i[6] == 1 && op->i[7] == 1 && op->i[8] == 1 && op->i[9] == 0 && op->i[10] == H && op->i[11] == W,
               "gemm_dma: pointwise layers only");
  OMNI_REQUIRE(op->f[0] == 0.0f, "gemm_dma: output scale is not supported");
  const long long M = (long long)B * H * W;
  OMNI_REQUIRE(M > 0 && M < (1ll << 31), "gemm_dma: bad M");
  a.M = (int)M;
  OMNI_REQUIRE(a.K >= 32 && a.K % 32 == 0 && a.N % 128 == 0, "gemm_dma: needs K %% 32 == 0 and N %% 128 == 0 (K %d, N %d)", a.K, a.N);
  OMNI_REQUIRE(a.ldi % 16 == 0 && a.in_coff % 16 == 0, "gemm_dma: split input needs 16-channel aligned ld / offset (%d, %d)", a.ldi, a.in_coff);
  OMNI_REQUIRE(a.ldo % 4 == 0 && a.out_coff % 4 == 0 && (!osplit || (a.ldo % 16 == 0 && a.out_coff % 16 == 0)), "gemm_dma: output alignment");
  OMNI_REQUIRE(!a.res || (a.ldr % 4 == 0 && a.res_coff % 4 == 0), "gemm_dma: residual alignment");
  OMNI_REQUIRE((long long)a.ldi * 4 * 256 < (1ll << 31) && (long long)a.K * 4 * 256 < (1ll << 31), "gemm_dma: row stride too large");
  a.nk = a.K / 32;
  // tile choice: 256x256 (one 8-wave block per CU, 128x64 per wave) when N allows, 256x128 otherwise — as long as the launch
  // still has a block for every CU; a short token matrix (small caption batches, 64x64 crops) takes 128x128 tiles instead, which
  // quarter the padded rows and give the chip 4x the blocks.  OMNI_GEMM_TILE (256x256 | 256x128 | 128x128) is the A/B knob of
  // tools/gemm_bench.py.
  int tile = (a.N % 256 == 0) ? 0 : 1;
  {
    const long long mt256 = (a.M + 255) / 256;
    if (mt256 * (a.N / (tile == 0 ? 256 : 128)) < 256) tile = (tile == 0 && mt256 * (a.N / 128) >= 256) ? 1 : 2;
  }
  if (const char* e = getenv("OMNI_GEMM_TILE")) {
    if (!strcmp(e, "256x128")) tile = 1;
    else if (!strcmp(e, "128x128")) tile = 2;
    else if (!strcmp(e, "256x256") && a.N % 256 == 0) tile = 0;
  }
  int rc;
  if (tile == 0) rc = launch_tile<256, 256, 2, 4, 2>(a, act, osplit, s);
  else if (tile == 1) rc = launch_tile<256, 128, 4, 2, 3>(a, act, osplit, s);
  else rc = launch_tile<128, 128, 2, 2, 2>(a, act, osplit, s);
  if (rc) return rc;
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

// OMNI_OP_SPLIT_CONVERT: f32 channel slice -> format B (in place allowed).
int omni_launch_split_convert(const omni_op_t* op, hipStream_t s) {
  const long long rows = (long long)op->i[0] * (op->i[1] > 0 ? op->i[1] : 1);
  const int C = op->i[3], ldi = op->i[4], in_coff = op->i[5], ldo = op->i[13], out_coff = op->i[14];
  OMNI_REQUIRE(op->dtype == OMNI_F32 && op->p[0] && op->p[4], "split_convert: f32 plans, non-null pointers");
  OMNI_REQUIRE(rows > 0 && C > 0 && C % 16 == 0 && ldi % 16 == 0 && in_coff % 16 == 0 && ldo % 16 == 0 && out_coff % 16 == 0,
               "split_convert: 16-channel alignment (C %d, ld %d/%d, off %d/%d)", C, ldi, ldo, in_coff, out_coff);
  const long long total = rows * (C / 16);
  hipLaunchKernelGGL(split_convert_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)op->p[0],
                     (unsigned char*)op->p[4], rows, C, ldi, in_coff, ldo, out_coff);
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

// OMNI_OP_MLP_FUSED (see include/omni_amd.h): y = res + fc2(GELU(fc1(x))) on format-B operands, C = 128.
int omni_launch_mlp_fused(const omni_op_t* op, hipStream_t s) {
  MlpArgs a{};
  a.x = (const unsigned char*)op->p[0]; a.w1 = (const unsigned char*)op->p[1]; a.b1 = (const float*)op->p[2];
  a.w2 = (const unsigned char*)op->p[5];
  a.ep.x = nullptr; a.ep.w = nullptr; a.ep.bias = (const float*)op->p[6]; a.ep.res = (const float*)op->p[3]; a.ep.y = (unsigned char*)op->p[4];
  const long long M = (long long)op->i[0] * (op->i[1] > 0 ? op->i[1] : 1);
  const int C = op->i[3], HID = op->i[12];
  a.ldi = op->i[4]; a.in_coff = op->i[5];
  a.ep.ldo = op->i[13]; a.ep.out_coff = op->i[14]; a.ep.ldr = op->i[16]; a.ep.res_coff = op->i[17];
  a.osc1 = op->f[1] != 0.0f ? op->f[1] : 1.0f;
  a.ep.oscale = op->f[2] != 0.0f ? op->f[2] : 1.0f;
  OMNI_REQUIRE(op->dtype == OMNI_F32, "mlp_fused: f32 plans only");
  OMNI_REQUIRE(a.x && a.w1 && a.w2 && a.b1 && a.ep.bias && a.ep.res && a.ep.y, "mlp_fused: null pointer (both biases and the residual are required)");
  OMNI_REQUIRE(C == 128 && HID == 512, "mlp_fused: built for C = 128, hidden = 512 (got %d, %d)", C, HID);
  OMNI_REQUIRE(M > 0 && M < (1ll << 31), "mlp_fused: bad row count");
  OMNI_REQUIRE(a.ldi % 16 == 0 && a.in_coff % 16 == 0, "mlp_fused: split input needs 16-channel aligned ld / offset");
  OMNI_REQUIRE(a.ep.ldo % 4 == 0 && a.ep.out_coff % 4 == 0 && a.ep.ldr % 4 == 0 && a.ep.res_coff % 4 == 0, "mlp_fused: output / residual alignment");
  a.ep.M = (int)M; a.ep.N = C; a.ep.K = HID; a.ep.nk = 0; a.ep.ldi = 0; a.ep.in_coff = 0;
  a.ep.mtiles = a.ep.ntiles = a.ep.xcd_order = a.ep.xcd_n = 0;
  hipLaunchKernelGGL((mlp_fused_kernel<128, 512>), dim3((unsigned)((M + 127) / 128)), dim3(256), 0, s, a);
  OMNI_HIP_CHECK(hipGetLastError());
  return OMNI_OK;
}

