// Shared by the GEMM kernels of libomni_amd.so (conv_igemm.hip, gemm_dma.hip): block -> tile order, activations.
#pragma once
#include "omni_internal.h"
#include <stdlib.h>

namespace {

// Block -> output tile.  MI355X dispatches consecutive workgroups round-robin over its 8 XCDs (bid & 7), each with
// a private 4 MiB L2.  With xcd_order the grid is laid out per XCD:
//   * xcd_n == 1: XCD x owns row blocks mt = x (mod 8) and walks all N tiles of one row block back to back, so the
//     activation tile is fetched once per row block and re-used from L2 by its N-tile neighbours;
//   * xcd_n  > 1: the N tiles are additionally partitioned over xcd_n XCD groups (XCD x serves N partition x % xcd_n
//     and row blocks = x / xcd_n (mod 8 / xcd_n)).  Each XCD then touches only ntiles/xcd_n weight panels — chosen so
//     that slab (<= 2 MiB) stays resident in its L2 while activations stream through — instead of the whole weight
//     matrix being re-fetched from Infinity Cache / HBM by every row block (measured in round 1: weight re-fetch was
//     ~65 % of this kernel's L2-miss traffic, profiles/r1_gemm_traffic_model.md).
// Pure index permutation: every (mt, nt) is produced exactly once for bid in [0, omni_tile_grid), results are
// bit-identical for any xcd_n.  Host mirror + exhaustive bijection test: omni_debug_tile_map / tests/test_host_cpu.py.
__host__ __device__ __forceinline__ bool tile_of_block(int bid, int mtiles, int ntiles, int xcd_order, int xcd_n, int& mt, int& nt) {
  if (xcd_order) {
    const int x = bid & 7, s = bid >> 3;
    const int gn = ntiles / xcd_n;               // N tiles per XCD group (xcd_n divides ntiles)
    const int ml = s / gn;
    mt = ml * (8 / xcd_n) + x / xcd_n;
    nt = (x % xcd_n) * gn + (s - ml * gn);
    return mt < mtiles;
  }
  mt = bid % mtiles;                             // few M tiles: plain order keeps all 8 XCDs busy
  nt = bid / mtiles;
  return true;
}

__host__ inline unsigned tile_grid(int mtiles, int ntiles, int xcd_order, int xcd_n) {
  if (!xcd_order) return (unsigned)(mtiles * ntiles);
  const int mper = 8 / xcd_n;
  return (unsigned)(((mtiles + mper - 1) / mper) * (ntiles / xcd_n) * 8);
}

// N-partition choice: smallest xcd_n in {2, 4, 8} dividing ntiles whose per-XCD weight slab fits the L2 budget;
// 1 (row-block mapping) when the whole matrix already fits or no divisor achieves residency.
__host__ inline int choose_xcd_n(int ntiles, long long weight_bytes) {
  const long long budget = 2ll << 20;            // half of the 4 MiB L2 (tools/l2_sim.py)
  if (weight_bytes <= budget) return 1;
  for (int xn = 2; xn <= 8; xn *= 2)
    if (ntiles % xn == 0 && weight_bytes / xn <= budget) return xn;
  return 1;
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == OMNI_ACT_SILU) {
    // torch CPU: x / (1 + exp(-x))
    return v / (1.0f + expf(-v));
  } else if (act == OMNI_ACT_GELU) {
    // exact erf GELU (hf ACT2FN["gelu"]): 0.5 * x * (1 + erf(x / sqrt(2)))
    return omni_gelu(v);
  }
  return v;
}

}  // namespace
