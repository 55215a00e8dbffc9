// Column sums as a ROLE: trailing workgroups of a launch (feta_lin_bwd) - or a launch of their own
// (feta_colsum_multi with segments of both shapes) - reduce a list of independent [R, C] buffers, one tile per
// workgroup of kColsumRoleThreads threads.  Two shapes of segment:
//   tall   16 columns x 16 row slices per workgroup, LDS tree over the slices (colsum_kernel's arithmetic)
//   wide   few rows x very many columns (split-K weight-gradient partials): one thread per 4 columns, every row
//          requested before the first add (colsum_wide_kernel's arithmetic)
// Deterministic: the order of the adds depends on the segment's shape only.
#pragma once
#include "feta_abi_common.h"
#include <feta_device.h>

namespace feta {

constexpr int kColsumRoleThreads = 256;

struct ColsumPlan {
  feta_colsum_seg seg[FETA_COLSUM_MAX_SEGS];
  int tile_end[FETA_COLSUM_MAX_SEGS];   // exclusive prefix end of each segment's tiles
  int wide[FETA_COLSUM_MAX_SEGS];
  int nseg;
};

inline bool colsum_seg_ok(const feta_colsum_seg& s) {
  return s.in && s.out && s.R > 0 && s.C > 0 && (s.ld == 0 || s.ld >= s.C) && (s.bcast_out == nullptr || s.bcast_rows > 0);
}

inline bool colsum_seg_wide(const feta_colsum_seg& s, int min_cols) {
  const int ld = s.ld > 0 ? s.ld : s.C;
  return s.R <= 512 && s.C >= min_cols && (s.C & 3) == 0 && (ld & 3) == 0 && aligned16(s.in) && aligned16(s.out) &&
         s.bcast_out == nullptr;
}

// fills the plan, returns the number of tiles (= workgroups of the role)
inline int plan_colsum(const feta_colsum_seg* segs, int nseg, ColsumPlan& p) {
  int tiles = 0;
  p.nseg = nseg;
  for (int i = 0; i < nseg; ++i) {
    const bool wide = colsum_seg_wide(segs[i], 1024);
    p.seg[i] = segs[i];
    p.wide[i] = wide ? 1 : 0;
    tiles += wide ? (segs[i].C / 4 + kColsumRoleThreads - 1) / kColsumRoleThreads : (segs[i].C + 15) / 16;
    p.tile_end[i] = tiles;
  }
  return tiles;
}

// out[c] = sum_r in[r][c] for one tile of one segment (256 threads): colsum_kernel's tree with 16 slices, or
// colsum_wide_kernel's row walk
__device__ __forceinline__ void colsum_role(const ColsumPlan& sg, int tile_id) {
  int si = 0;
  while (si + 1 < sg.nseg && tile_id >= sg.tile_end[si]) ++si;
  const feta_colsum_seg s = sg.seg[si];
  const int tile = tile_id - (si > 0 ? sg.tile_end[si - 1] : 0);
  const int ld = s.ld > 0 ? s.ld : s.C;
  if (sg.wide[si]) {
    const int c4 = tile * kColsumRoleThreads + (int)threadIdx.x;
    if (c4 >= s.C / 4) return;
    const float* p = s.in + 4 * (int64_t)c4;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int r = 0;
    for (; r + 8 <= s.R; r += 8) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(p + (int64_t)(r + i) * ld);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w;
      }
    }
    for (; r < s.R; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)r * ld);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(s.out + 4 * (int64_t)c4) = acc;
    return;
  }
  constexpr int COLS = 16, SL = kColsumRoleThreads / COLS;
  float* red = feta_lds;   // [SL][COLS]
  const int lc = threadIdx.x & (COLS - 1), slice = threadIdx.x / COLS;
  const int col = tile * COLS + lc;
  float acc = 0.0f;
  if (col < s.C)
    for (int r = slice; r < s.R; r += SL) acc += s.in[(int64_t)r * ld + col];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int half = SL / 2; half >= 1; half >>= 1) {
    if (slice < half) red[threadIdx.x] += red[threadIdx.x + half * COLS];
    __syncthreads();
  }
  if (col < s.C) {
    const float v = red[lc];
    if (slice == 0) s.out[col] = v;
    if (s.bcast_out != nullptr)
      for (int r = slice; r < s.bcast_rows; r += SL) s.bcast_out[(int64_t)r * s.C + col] = v;
  }
}

}  // namespace feta
