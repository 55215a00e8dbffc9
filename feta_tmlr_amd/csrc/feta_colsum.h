// Column sums as a ROLE: trailing workgroups of a launch (feta_lin_bwd) - or a launch of their own
// (feta_colsum_multi with segments of both shapes) - reduce a list of independent [R, C] buffers, one tile per
// workgroup of kColsumRoleThreads threads.  Two shapes of segment:
//   tall   16 columns x (threads / 16) row slices per workgroup, LDS tree over the slices
//   wide   few rows x very many columns (split-K weight-gradient partials): one thread per 4 columns (256 float4
//          columns per workgroup, larger workgroups split the rows), eight rows requested before the first add
// Deterministic: the order of the adds depends on the segment's shape and the workgroup size only.
#pragma once
#include <cstdlib>

#include "feta_abi_common.h"
#include <feta_device.h>

namespace feta {

constexpr int kColsumRoleThreads = 256;    // as a role of feta_lin_bwd
constexpr int kColsumMixedThreads = 1024;  // as a launch of its own

struct ColsumPlan {
  feta_colsum_seg seg[FETA_COLSUM_MAX_SEGS];
  int tile_end[FETA_COLSUM_MAX_SEGS];   // exclusive prefix end of each segment's tiles
  int wide[FETA_COLSUM_MAX_SEGS];
  int nseg;
  int wq;   // float4 columns of a wide tile (plan_colsum)
};

inline bool colsum_seg_ok(const feta_colsum_seg& s) {
  return s.in && s.out && s.R > 0 && s.C > 0 && (s.ld == 0 || s.ld >= s.C) && (s.bcast_out == nullptr || s.bcast_rows > 0);
}

inline bool colsum_seg_wide(const feta_colsum_seg& s, int min_cols) {
  const int ld = s.ld > 0 ? s.ld : s.C;
  return s.R <= 512 && s.C >= min_cols && (s.C & 3) == 0 && (ld & 3) == 0 && aligned16(s.in) && aligned16(s.out) &&
         s.bcast_out == nullptr;
}

// float4 columns of a wide tile for a workgroup of `threads` threads: 256 (one thread per float4 column, larger
// workgroups split the rows in 2) - but 64 for the 1024-thread launch of its own, whose segments are few rows x ~10^4
// columns (the stack's split-K partials: 128 rows x 16 640 columns were 17 workgroups of 32 rows per thread, four dependent
// batches; 65 workgroups of 8 rows per thread are one batch and fill a quarter of the chip instead of a sixteenth)
constexpr int colsum_wide_q(int threads) { return threads >= 1024 ? 64 : 256; }
inline int colsum_wide_adapt() {   // FETA_COLSUM_ADAPT=0: always the tile width above (A/B timing)
  const char* e = getenv("FETA_COLSUM_ADAPT");
  return (e != nullptr && e[0] == '0') ? 0 : 1;
}

// fills the plan, returns the number of tiles (= workgroups of the role); `threads`: the workgroup size the tiles will
// be reduced with (colsum_role<THREADS>)
inline int plan_colsum(const feta_colsum_seg* segs, int nseg, ColsumPlan& p, int threads = 256) {
  int wq = colsum_wide_q(threads);
  p.nseg = nseg;
  for (;;) {
    int tiles = 0;
    for (int i = 0; i < nseg; ++i) {
      const bool wide = colsum_seg_wide(segs[i], 4096);
      p.seg[i] = segs[i];
      p.wide[i] = wide ? 1 : 0;
      tiles += wide ? (segs[i].C / 4 + wq - 1) / wq : (segs[i].C + 15) / 16;
      p.tile_end[i] = tiles;
    }
    p.wq = wq;
    // the 1024-thread launch holds two workgroups per CU (512 at a time): where 64-column tiles are more than that - the
    // partials of configs 4 / 5: 616 - 641 tiles, a second round for a fifth of them - tiles of twice the columns
    if (threads < 1024 || tiles <= 512 || wq >= 256 || colsum_wide_adapt() == 0) return tiles;
    wq *= 2;
  }
}

// dynamic LDS floats the role needs in a workgroup of THREADS threads
constexpr int colsum_role_lds_floats(int threads) { return threads > 256 ? 4 * threads : threads; }

__device__ __forceinline__ void add4(float4& a, const float4& v) {
  a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
}

// out[c] = sum_r in[r][c] for one tile of one segment, by a workgroup of THREADS (256 or 1024) threads; every thread
// of the workgroup must call it (barriers).  The order of the adds depends on the shape and on THREADS only.
template <int THREADS>
__device__ __forceinline__ void colsum_role(const ColsumPlan& sg, int tile_id) {
  int si = 0;
  while (si + 1 < sg.nseg && tile_id >= sg.tile_end[si]) ++si;
  const feta_colsum_seg s = sg.seg[si];
  const int tile = tile_id - (si > 0 ? sg.tile_end[si - 1] : 0);
  const int ld = s.ld > 0 ? s.ld : s.C;
  const int tid = threadIdx.x;
  if (sg.wide[si]) {
    // WQ float4 columns per tile; THREADS / WQ row slices, tree over the slices through LDS
    const int WQ = sg.wq, SLW = THREADS / WQ;
    const int c4 = tile * WQ + (tid % WQ), slice = tid / WQ;
    const bool ok = c4 < s.C / 4;
    const float* p = s.in + 4 * (int64_t)(ok ? c4 : 0);
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int r = slice;
    for (; r + 7 * SLW < s.R; r += 8 * SLW) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(p + (int64_t)(r + i * SLW) * ld);
#pragma unroll
      for (int i = 0; i < 8; ++i) add4(acc, v[i]);
    }
    for (; r < s.R; r += SLW) add4(acc, *reinterpret_cast<const float4*>(p + (int64_t)r * ld));
    if (SLW > 1) {
      float4* red = reinterpret_cast<float4*>(feta_lds);   // [SLW][WQ]
      red[tid] = acc;
      __syncthreads();
      for (int half = SLW / 2; half >= 1; half >>= 1) {
        if (slice < half) add4(red[tid], red[tid + half * WQ]);
        __syncthreads();
      }
      acc = red[tid % WQ];
    }
    if (ok && slice == 0) *reinterpret_cast<float4*>(s.out + 4 * (int64_t)c4) = acc;
    return;
  }
  // 16 columns x THREADS / 16 row slices, pairwise tree over the slices
  constexpr int COLS = 16, SL = THREADS / COLS;
  float* red = feta_lds;   // [SL][COLS]
  const int lc = tid & (COLS - 1), slice = tid / COLS;
  const int col = tile * COLS + lc;
  float acc = 0.0f;
  if (col < s.C) {
    const float* p = s.in + col;
    int r = slice;
    for (; r + 3 * SL < s.R; r += 4 * SL) {   // four rows in flight
      const float v0 = p[(int64_t)r * ld], v1 = p[(int64_t)(r + SL) * ld], v2 = p[(int64_t)(r + 2 * SL) * ld],
                  v3 = p[(int64_t)(r + 3 * SL) * ld];
      acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; r < s.R; r += SL) acc += p[(int64_t)r * ld];
  }
  red[tid] = acc;
  __syncthreads();
  for (int half = SL / 2; half >= 1; half >>= 1) {
    if (slice < half) red[tid] += red[tid + half * COLS];
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
