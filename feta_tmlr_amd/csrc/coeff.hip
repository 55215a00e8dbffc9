// A2: attention-conditioned filter coefficients.  Replaces the body of
// DiffTransformerEncoderGenGCN.get_filter_coefficients (transformer/models.py:240-283)
// up to the mean pool: the reference materialises a dense H*sum(n_b^2)-edge graph per
// batch on the host and runs GCNConv(C,C) on an all-ones [H*N_tot, C] input.  Because the
// input is all ones, GCNConv(ones)[j] = c_j * colsum(W) + bias exactly, with c_j the sum of
// the GCN-normalised weights into node j (normalisation text: transformer/GenGCN.py:55-102).
// One workgroup per (head, graph) block:
//   stage attn[b,h,:n,:] in LDS (one coalesced sweep of the only HBM read)
//   deg_j = sum_i w_ij ; c_j = sum_i deg_i^-1/2 w_ij deg_j^-1/2      (two LDS column sweeps)
//   pooled[c] = mean_j tanh(c_j s_c + b_c)                          (thread per channel)
#include <cmath>

#include "feta_abi_common.h"
#include <feta_device.h>
#include <cstdlib>

#include "feta_coeff.h"
#include "feta_colsum.h"

namespace feta {

__global__ __launch_bounds__(kCoeffThreads) void coeff_fwd_kernel(
    const float* __restrict__ attn, const int32_t* __restrict__ n_real, const float* __restrict__ s,
    const float* __restrict__ gbias, float* __restrict__ cj_out, float* __restrict__ pooled, int B,
    int N, int H, int C, int stage) {
  coeff_fwd_body(attn, n_real, s, gbias, cj_out, pooled, B, N, H, C, stage, (int)blockIdx.x, (int)blockIdx.y,
                 (int)gridDim.y);
}

// graphs beyond 64 nodes (config 4): ONE workgroup of 1024 threads per block.  The 256-thread form ran four channel-slice
// workgroups per block, each staging the block's N x N attention rows again (two rounds of requests) and then walking the
// nodes with two waves per SIMD: 23 us at B = 64, N = 128.  Here the rows arrive in one round (16 bytes x 4 per thread),
// the column sweeps have 32 row slices, and every thread owns one of the 1024 channels with four waves per SIMD under the
// tanh chain.
__global__ __launch_bounds__(kCoeffWideThreads) void coeff_fwd_wide_kernel(
    const float* __restrict__ attn, const int32_t* __restrict__ n_real, const float* __restrict__ s,
    const float* __restrict__ gbias, float* __restrict__ cj_out, float* __restrict__ pooled, int B,
    int N, int H, int C) {
  coeff_fwd_body<kCoeffWideThreads>(attn, n_real, s, gbias, cj_out, pooled, B, N, H, C, 2, (int)blockIdx.x);
}

__global__ __launch_bounds__(kCoeffThreads) void coeff_bwd_kernel(
    const float* __restrict__ cj, const int32_t* __restrict__ n_real, const float* __restrict__ s,
    const float* __restrict__ gbias, const float* __restrict__ dpooled, float* __restrict__ partial,
    int B, int N, int H, int C, int G) {
  coeff_bwd_body(cj, n_real, s, gbias, dpooled, partial, B, N, H, C, G, (int)blockIdx.x, (int)blockIdx.y);
}

// out[c] = sum_r in[r][c]: 16 columns x 64 row slices per workgroup (64-byte row segments),
// pairwise LDS tree over the slices; deterministic.
constexpr int kCsCols = 16, kCsSlices = 64;
__global__ __launch_bounds__(kCsCols * kCsSlices) void colsum_kernel(const float* __restrict__ in,
                                                                      float* __restrict__ out, int R,
                                                                      int C, int ld) {
  float* red = feta_lds;  // [slices][cols]
  const int lc = threadIdx.x & (kCsCols - 1);
  const int slice = threadIdx.x / kCsCols;
  const int col = blockIdx.x * kCsCols + lc;
  float acc = 0.0f;
  if (col < C)
    for (int r = slice; r < R; r += kCsSlices) acc += in[(int64_t)r * ld + col];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int half = kCsSlices / 2; half >= 1; half >>= 1) {
    if (slice < half) red[threadIdx.x] += red[threadIdx.x + half * kCsCols];
    __syncthreads();
  }
  if (slice == 0 && col < C) out[col] = red[lc];
}

// few rows x very many columns (the split-K weight-gradient partials of a whole layer stack:
// ~20 x 100k): one thread per 4 columns, every row requested before the first add
__global__ __launch_bounds__(64) void colsum_wide_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         int R, int C4, int ld) {
  const int c4 = blockIdx.x * 64 + threadIdx.x;
  if (c4 >= C4) return;
  const float* p = in + 4 * (int64_t)c4;
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  int r = 0;
  for (; r + 8 <= R; r += 8) {
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(p + (int64_t)(r + i) * ld);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w;
    }
  }
  for (; r < R; ++r) {
    const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)r * ld);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(out + 4 * (int64_t)c4) = acc;
}

int launch_colsum_strided(const float* in, float* out, int R, int C, int ld, hipStream_t stream) {
  if (R <= 256 && C >= 4096 && (C & 3) == 0 && (ld & 3) == 0 && aligned16(in) && aligned16(out)) {
    const int c4 = C / 4;
    auto kern = colsum_wide_kernel;
    hipLaunchKernelGGL(kern, dim3((c4 + 63) / 64), dim3(64), 0, stream, in, out, R, c4, ld);
    return check_launch("feta_colsum");
  }
  const dim3 grid((C + kCsCols - 1) / kCsCols), block(kCsCols * kCsSlices);
  auto kern = colsum_kernel;
  hipLaunchKernelGGL(kern, grid, block, kCsCols * kCsSlices * sizeof(float), stream, in, out, R, C, ld);
  return check_launch("feta_colsum");
}
int launch_colsum(const float* in, float* out, int R, int C, hipStream_t stream) {
  return launch_colsum_strided(in, out, R, C, C, stream);
}

// Several independent column sums in ONE launch (every launch of a captured step costs ~4.5 us, whatever
// its size): workgroup -> (segment, 16-column tile) through the prefix of tile counts; same tree as
// colsum_kernel.  A segment may also write its result to every row of a dense [bcast_rows][C] matrix: the
// gradient of a parameter that only ever multiplies an all-ones input (every row equal).
struct ColsumSegs {
  feta_colsum_seg seg[FETA_COLSUM_MAX_SEGS];
  int tile_end[FETA_COLSUM_MAX_SEGS];   // exclusive prefix end of each segment's tiles
  int nseg;
};

__global__ __launch_bounds__(kCsCols * kCsSlices) void colsum_multi_kernel(ColsumSegs a) {
  float* red = feta_lds;  // [slices][cols]
  int si = 0;
  while (si + 1 < a.nseg && (int)blockIdx.x >= a.tile_end[si]) ++si;
  const feta_colsum_seg sg = a.seg[si];
  const int tile = (int)blockIdx.x - (si > 0 ? a.tile_end[si - 1] : 0);
  const int lc = threadIdx.x & (kCsCols - 1);
  const int slice = threadIdx.x / kCsCols;
  const int col = tile * kCsCols + lc;
  const int ld = sg.ld > 0 ? sg.ld : sg.C;
  float acc = 0.0f;
  if (col < sg.C)
    for (int r = slice; r < sg.R; r += kCsSlices) acc += sg.in[(int64_t)r * ld + col];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int half = kCsSlices / 2; half >= 1; half >>= 1) {
    if (slice < half) red[threadIdx.x] += red[threadIdx.x + half * kCsCols];
    __syncthreads();
  }
  if (col < sg.C) {
    const float v = red[lc];
    if (slice == 0) sg.out[col] = v;
    if (sg.bcast_out != nullptr)
      for (int r = slice; r < sg.bcast_rows; r += kCsSlices) sg.bcast_out[(int64_t)r * sg.C + col] = v;
  }
}

// the same for segments of few rows x very many columns (the split-K weight-gradient partials of the layer stack: one
// buffer per partial-row count): colsum_wide_kernel's arithmetic, workgroup -> (segment, 64 float4 columns)
__global__ __launch_bounds__(64) void colsum_wide_multi_kernel(ColsumSegs a) {
  int si = 0;
  while (si + 1 < a.nseg && (int)blockIdx.x >= a.tile_end[si]) ++si;
  const feta_colsum_seg sg = a.seg[si];
  const int c4 = ((int)blockIdx.x - (si > 0 ? a.tile_end[si - 1] : 0)) * 64 + threadIdx.x;
  if (c4 >= sg.C / 4) return;
  const int ld = sg.ld > 0 ? sg.ld : sg.C;
  const float* p = sg.in + 4 * (int64_t)c4;
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  int r = 0;
  for (; r + 8 <= sg.R; r += 8) {
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(p + (int64_t)(r + i) * ld);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w;
    }
  }
  for (; r < sg.R; ++r) {
    const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)r * ld);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(sg.out + 4 * (int64_t)c4) = acc;
}

// segments of both shapes in one launch (feta_colsum.h)
__global__ __launch_bounds__(kColsumMixedThreads) void colsum_mixed_kernel(ColsumPlan p) {
  colsum_role<kColsumMixedThreads>(p, (int)blockIdx.x);
}

int launch_colsum_multi(const feta_colsum_seg* segs, int nseg, hipStream_t stream) {
  ColsumSegs a{};
  a.nseg = nseg;
  bool wide = true, any_wide = false;
  for (int i = 0; i < nseg; ++i) {
    const bool w = colsum_seg_wide(segs[i], 4096);
    wide = wide && w;
    any_wide = any_wide || w;
  }
  if (any_wide && !wide) {
    ColsumPlan p{};
    const int tiles = plan_colsum(segs, nseg, p, kColsumMixedThreads);
    auto kern = colsum_mixed_kernel;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(kColsumMixedThreads),
                       colsum_role_lds_floats(kColsumMixedThreads) * sizeof(float), stream, p);
    return check_launch("feta_colsum_multi");
  }
  if (wide) {
    int tiles = 0;
    for (int i = 0; i < nseg; ++i) {
      a.seg[i] = segs[i];
      tiles += (segs[i].C / 4 + 63) / 64;
      a.tile_end[i] = tiles;
    }
    auto kern = colsum_wide_multi_kernel;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64), 0, stream, a);
    return check_launch("feta_colsum_multi");
  }
  int tiles = 0;
  for (int i = 0; i < nseg; ++i) {
    a.seg[i] = segs[i];
    tiles += (segs[i].C + kCsCols - 1) / kCsCols;
    a.tile_end[i] = tiles;
  }
  auto kern = colsum_multi_kernel;
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(kCsCols * kCsSlices), kCsCols * kCsSlices * sizeof(float), stream, a);
  return check_launch("feta_colsum_multi");
}

}  // namespace feta

using namespace feta;

extern "C" int feta_coeff_fwd(const float* attn, const int32_t* n_real, const float* s,
                              const float* gcn_bias, float* cj, float* pooled, int B, int N, int H,
                              int C, feta_stream_t stream) {
  FETA_REQUIRE(attn && n_real && s && gcn_bias && cj && pooled, "coeff_fwd: null pointer");
  FETA_REQUIRE(B > 0 && H > 0 && C > 0 && N > 0 && N <= kCoeffThreads,
               "coeff_fwd: need 0 < N <= %d (got %d)", kCoeffThreads, N);
  // stand-alone launch: the block's attention rows are staged whenever they fit a CU's LDS (N <= 192: the unstaged
  // column sweeps are 2 N dependent trips to L2 - 57 us at N = 128), and graphs beyond 64 nodes split the channels of a
  // block over 4 workgroups (config 4, PATTERN)
  const int want = N > 64 ? 2 : 1;
  const int stage = sizeof(float) * (size_t)coeff_fwd_lds_floats_mode(N, want) <= 150 * 1024 ? want : 0;
  const size_t lds = sizeof(float) * (size_t)coeff_fwd_lds_floats_mode(N, stage);
  if (N > 64) {
    const size_t wide = sizeof(float) * (size_t)coeff_fwd_lds_floats_mode(N, 2, kCoeffWideThreads);
    const char* e = getenv("FETA_COEFF_WIDE");
    if (wide <= 160 * 1024 && !(e != nullptr && e[0] == '0')) {
      auto kern = coeff_fwd_wide_kernel;
      static LdsSeen wide_seen;
      allow_dynamic_lds(kern, wide, wide_seen);
      hipLaunchKernelGGL(kern, dim3(B * H), dim3(kCoeffWideThreads), wide, (hipStream_t)stream, attn, n_real, s, gcn_bias,
                         cj, pooled, B, N, H, C);
      return check_launch("feta_coeff_fwd");
    }
  }
  const int cs = (N > 64 && C >= 4 * kCoeffThreads) ? 4 : 1;
  const dim3 grid(B * H, cs), block(kCoeffThreads);
  auto kern = coeff_fwd_kernel;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, attn, n_real, s, gcn_bias, cj,
                     pooled, B, N, H, C, stage);
  return check_launch("feta_coeff_fwd");
}

extern "C" int feta_coeff_bwd_groups(int B, int H) {
  const int t = B * H;
  // 128 groups x C / 256 channel tiles fill the chip twice at C = 1024 while a group walks a few blocks; from 2048 blocks on
  // (config 5: 4096) a group of 128 walked 32 blocks with two waves per SIMD - 58.7 us stand-alone against a ~15 us tanh
  // floor - so the groups grow with the batch (4 x the partial rows for the reduction launch)
  if (t >= 2048) return 4 * kCoeffGroupsMax;
  return t < kCoeffGroupsMax ? t : kCoeffGroupsMax;
}

extern "C" int feta_colsum_multi(const feta_colsum_seg* segs, int nseg, feta_stream_t stream) {
  FETA_REQUIRE(segs != nullptr && nseg >= 1 && nseg <= FETA_COLSUM_MAX_SEGS, "colsum_multi: 1..%d segments",
               FETA_COLSUM_MAX_SEGS);
  for (int i = 0; i < nseg; ++i)
    FETA_REQUIRE(segs[i].in && segs[i].out && segs[i].R > 0 && segs[i].C > 0 && (segs[i].ld == 0 || segs[i].ld >= segs[i].C) &&
                 (segs[i].bcast_out == nullptr || segs[i].bcast_rows > 0), "colsum_multi: bad segment %d", i);
  return launch_colsum_multi(segs, nseg, (hipStream_t)stream);
}

extern "C" int feta_coeff_bwd(const float* cj, const int32_t* n_real, const float* s,
                              const float* gcn_bias, const float* dpooled, float* partial, float* ds,
                              float* dbias, float* dw_dense, int dw_rows, int B, int N, int H, int C,
                              feta_stream_t stream) {
  FETA_REQUIRE(cj && n_real && s && gcn_bias && dpooled && partial && (ds == nullptr || dbias != nullptr),
               "coeff_bwd: null pointer");
  FETA_REQUIRE(B > 0 && H > 0 && C > 0 && N > 0, "coeff_bwd: empty shape");
  const int G = feta_coeff_bwd_groups(B, H);
  const dim3 grid((C + kCoeffThreads - 1) / kCoeffThreads, G), block(kCoeffThreads);
  auto kern = coeff_bwd_kernel;
  hipLaunchKernelGGL(kern, grid, block, kCoeffPass * N * sizeof(float), (hipStream_t)stream, cj, n_real, s,
                     gcn_bias, dpooled, partial, B, N, H, C, G);
  int rc = check_launch("feta_coeff_bwd");
  if (rc != FETA_OK || ds == nullptr) return rc;   // ds == NULL: the caller reduces the [G, 2C] partials itself
  if (dw_dense != nullptr) {
    // one launch: ds (also written to every row of the dense [dw_rows][C] weight gradient) and dbias
    FETA_REQUIRE(dw_rows > 0, "coeff_bwd: dw_dense needs dw_rows");
    feta_colsum_seg segs[2] = {{partial, ds, G, C, 2 * C, dw_dense, dw_rows}, {partial + C, dbias, G, C, 2 * C, nullptr, 0}};
    return launch_colsum_multi(segs, 2, (hipStream_t)stream);
  }
  if (dbias == ds + C)  // contiguous outputs: one reduction launch for both
    return launch_colsum(partial, ds, G, 2 * C, (hipStream_t)stream);
  // separate outputs: reduce the two interleaved halves one after the other
  rc = launch_colsum_strided(partial, ds, G, C, 2 * C, (hipStream_t)stream);
  if (rc != FETA_OK) return rc;
  return launch_colsum_strided(partial + C, dbias, G, C, 2 * C, (hipStream_t)stream);
}

extern "C" int feta_colsum(const float* in, float* out, int R, int C, feta_stream_t stream) {
  FETA_REQUIRE(in && out && R > 0 && C > 0, "colsum: bad arguments");
  return launch_colsum(in, out, R, C, (hipStream_t)stream);
}
