// The C x C ``self.linear`` of the coefficient generator (transformer/models.py:284: coeff = pooled W^T + b on the
// [H*B, C] pooled rows) - forward as one launch, backward (dX, dW, db) as ONE launch.
//
// At the BASELINE batch the three products are 512 x 512 x 512 fp32 GEMMs: 0.27 GFLOP each, 1.7 us of the fp32 matrix
// pipe if all 1024 SIMDs take part.  The library kernels the framework picks for them (64 x 64 macro tiles: 64
// workgroups on 256 CUs) take ~13 us each, three launches; here every wave owns ONE 16 x 16 output tile over the whole
// contraction (1024 waves for a 512 x 512 result: one per SIMD), operands come straight from L2 (3 MB in all, no LDS
// stage: a wave's 16-row operand slab is read exactly once by it), and the two gradient products share a launch -
// workgroups take roles - together with the bias gradient (row sums of the dW role's own operand) and any column sums
// the caller has pending (the filter-bias partials, linear_cat's split-K partials): launches, not flops, are what a
// captured step at this batch pays for (~4.5 us each).
//
// Arithmetic: v_mfma_f32_16x16x4_f32, k-ordered exact fp32 - the same contraction order for every output element
// (k ascending in steps of 16, inside a step the four k-quads of the MFMA), deterministic.
#include "feta_abi_common.h"
#include "feta_colsum.h"
#include "feta_tiles.h"

namespace feta {

constexpr int kLinThreads = kColsumRoleThreads;   // 4 waves = a 32 x 32 output tile

// One wave: acc[r] = sum_k A(m0 + 4g + r, k) B(k, n0 + lq).
//   A_KC: A stored [row][k] (k contiguous: one float4 per step) else [k][row] (four scalar loads, lanes contiguous)
//   B_KC: B stored [col][k] else [k][col]
// rowA / colB are this lane's (clamped) row of A and column of B; Kc % 4 == 0.
template <bool A_KC, bool B_KC>
__device__ __forceinline__ f32x4 wave_gemm_tile(const float* __restrict__ A, int64_t lda, int rowA,
                                                 const float* __restrict__ B, int64_t ldb, int colB, int Kc, int g,
                                                 float* rowsum) {
  f32x4 acc = zero4();
  float rs = 0.0f;
  const float* ap = A_KC ? A + (int64_t)rowA * lda : A + rowA;
  const float* bp = B_KC ? B + (int64_t)colB * ldb : B + colB;
  const int full = Kc / 16;
#pragma unroll 4
  for (int j = 0; j < full; ++j) {
    const int k0 = 16 * j + 4 * g;
    float a[4], b[4];
    if (A_KC) {
      const float4 v = *reinterpret_cast<const float4*>(ap + k0);
      a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = ap[(int64_t)(k0 + s) * lda];
    }
    if (B_KC) {
      const float4 v = *reinterpret_cast<const float4*>(bp + k0);
      b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) b[s] = bp[(int64_t)(k0 + s) * ldb];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc = mfma16(a[s], b[s], acc);
      rs += a[s];
    }
  }
  if (16 * full < Kc) {   // ragged end of the contraction: quads beyond Kc contribute zeros
    const int k0 = 16 * full + 4 * g;
    const bool ok = k0 < Kc;
    const int kk = ok ? k0 : 0;
    float a[4], b[4];
    if (A_KC) {
      const float4 v = *reinterpret_cast<const float4*>(ap + kk);
      a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = ap[(int64_t)(kk + s) * lda];
    }
    if (B_KC) {
      const float4 v = *reinterpret_cast<const float4*>(bp + kk);
      b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) b[s] = bp[(int64_t)(kk + s) * ldb];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float av = ok ? a[s] : 0.0f, bv = ok ? b[s] : 0.0f;
      acc = mfma16(av, bv, acc);
      rs += av;
    }
  }
  if (rowsum != nullptr) *rowsum = rs;
  return acc;
}

// D[m][n] (+ bias[n]) for this wave's tile; rows >= M / columns >= N are not stored
__device__ __forceinline__ void store_tile(float* __restrict__ D, int64_t ldd, int m0, int n0, int M, int N,
                                           const f32x4& acc, const float* __restrict__ bias, int lq, int g) {
  const int col = n0 + lq;
  if (col >= N) return;
  const float bv = bias != nullptr ? bias[col] : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 4 * g + r;
    if (row < M) D[(int64_t)row * ldd + col] = acc[r] + bv;
  }
}

struct LinFwdArgs {
  const float* x;     // [R][K]
  const float* w;     // [N][K]
  const float* bias;  // [N], nullable
  float* y;           // [R][N]
  int R, K, N;
  int tn;             // 32-column tiles of y
};

__global__ __launch_bounds__(kLinThreads) void lin_fwd_kernel(LinFwdArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lq = lane & 15, g = lane >> 4;
  const int tm = (int)blockIdx.x / a.tn, tn = (int)blockIdx.x - tm * a.tn;
  const int m0 = 32 * tm + 16 * (wv & 1), n0 = 32 * tn + 16 * (wv >> 1);
  if (m0 >= a.R || n0 >= a.N) return;
  const f32x4 acc = wave_gemm_tile<true, true>(a.x, a.K, min(m0 + lq, a.R - 1), a.w, a.K, min(n0 + lq, a.N - 1), a.K,
                                               g, nullptr);
  store_tile(a.y, a.N, m0, n0, a.R, a.N, acc, a.bias, lq, g);
}

struct LinBwdArgs {
  const float* x;    // [R][K]  (pooled)
  const float* w;    // [N][K]
  const float* dy;   // [R][N]  (dcoeff)
  float* dx;         // [R][K], nullable
  float* dw;         // [N][K]
  float* db;         // [N], nullable
  int R, K, N;
  int tk;            // 32-column tiles over K (both products have K columns)
  int nx, nw;        // workgroups of the dX role / of the dW role
  ColsumPlan segs;
};

__global__ __launch_bounds__(kLinThreads) void lin_bwd_kernel(LinBwdArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lq = lane & 15, g = lane >> 4;
  int blk = (int)blockIdx.x;
  if (blk < a.nx) {
    // dx[r][k] = sum_o dy[r][o] w[o][k]
    const int tm = blk / a.tk, tn = blk - tm * a.tk;
    const int m0 = 32 * tm + 16 * (wv & 1), n0 = 32 * tn + 16 * (wv >> 1);
    if (m0 >= a.R || n0 >= a.K) return;
    const f32x4 acc = wave_gemm_tile<true, false>(a.dy, a.N, min(m0 + lq, a.R - 1), a.w, a.K, min(n0 + lq, a.K - 1),
                                                  a.N, g, nullptr);
    store_tile(a.dx, a.K, m0, n0, a.R, a.K, acc, nullptr, lq, g);
    return;
  }
  blk -= a.nx;
  if (blk < a.nw) {
    // dw[o][k] = sum_r dy[r][o] x[r][k];  db[o] = sum_r dy[r][o] (the row sums of this role's A operand)
    const int tm = blk / a.tk, tn = blk - tm * a.tk;
    const int m0 = 32 * tm + 16 * (wv & 1), n0 = 32 * tn + 16 * (wv >> 1);
    if (m0 >= a.N || n0 >= a.K) return;
    float rs = 0.0f;
    const f32x4 acc = wave_gemm_tile<false, false>(a.dy, a.N, min(m0 + lq, a.N - 1), a.x, a.K, min(n0 + lq, a.K - 1),
                                                   a.R, g, &rs);
    store_tile(a.dw, a.K, m0, n0, a.N, a.K, acc, nullptr, lq, g);
    if (a.db != nullptr && n0 == 0) {
      rs += shfl_xor(rs, 16);
      rs += shfl_xor(rs, 32);
      if (g == 0 && m0 + lq < a.N) a.db[m0 + lq] = rs;
    }
    return;
  }
  colsum_role<kLinThreads>(a.segs, blk - a.nw);
}

}  // namespace feta

using namespace feta;

extern "C" int feta_lin_supported(int R, int K, int N) {
  return (R >= 1 && K >= 16 && N >= 16 && (K & 3) == 0 && (N & 3) == 0 && (R & 3) == 0) ? 1 : 0;
}

extern "C" int feta_lin_fwd(const float* x, const float* w, const float* bias, float* y, int R, int K, int N,
                            feta_stream_t stream) {
  FETA_REQUIRE(x && w && y, "lin_fwd: null pointer");
  FETA_REQUIRE(feta_lin_supported(R, K, N), "lin_fwd: need R, K, N multiples of 4, K, N >= 16 (R=%d K=%d N=%d)", R, K, N);
  FETA_REQUIRE(aligned16(x) && aligned16(w), "lin_fwd: x and w must be 16-byte aligned");
  LinFwdArgs a{x, w, bias, y, R, K, N, (N + 31) / 32};
  const int grid = ((R + 31) / 32) * a.tn;
  auto kern = lin_fwd_kernel;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kLinThreads), 0, (hipStream_t)stream, a);
  return check_launch("feta_lin_fwd");
}

extern "C" int feta_lin_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int R,
                            int K, int N, const feta_colsum_seg* segs, int nseg, feta_stream_t stream) {
  FETA_REQUIRE(x && w && dy && dw, "lin_bwd: null pointer");
  FETA_REQUIRE(feta_lin_supported(R, K, N), "lin_bwd: need R, K, N multiples of 4, K, N >= 16 (R=%d K=%d N=%d)", R, K, N);
  FETA_REQUIRE(aligned16(dy), "lin_bwd: dy must be 16-byte aligned");
  FETA_REQUIRE(nseg >= 0 && nseg <= FETA_COLSUM_MAX_SEGS && (nseg == 0 || segs != nullptr),
               "lin_bwd: 0..%d column-sum segments", FETA_COLSUM_MAX_SEGS);
  LinBwdArgs a{};
  a.x = x; a.w = w; a.dy = dy; a.dx = dx; a.dw = dw; a.db = db;
  a.R = R; a.K = K; a.N = N;
  a.tk = (K + 31) / 32;
  a.nx = dx != nullptr ? ((R + 31) / 32) * a.tk : 0;
  a.nw = ((N + 31) / 32) * a.tk;
  for (int i = 0; i < nseg; ++i) FETA_REQUIRE(colsum_seg_ok(segs[i]), "lin_bwd: bad segment %d", i);
  const int tiles = plan_colsum(segs, nseg, a.segs);
  auto kern = lin_bwd_kernel;
  hipLaunchKernelGGL(kern, dim3(a.nx + a.nw + tiles), dim3(kLinThreads), kLinThreads * sizeof(float),
                     (hipStream_t)stream, a);
  return check_launch("feta_lin_bwd");
}
