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
#include <cstdlib>

#include "feta_abi_common.h"
#include "feta_colsum.h"
#include "feta_lp.h"

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


// ---- LDS-tiled form for shapes whose dims are multiples of 64 (the BASELINE shape: R = H*B = 512, K = N = C = 1024, one
// GFLOP per product - compute-bound, where the one-tile-per-wave form above is L2-bound: every wave re-reads a whole
// operand slab).  D[i][j] = sum_kk P[i][kk] Q[kk][j], 64 x 32 output tile per workgroup, 4 waves = (32-row half, 16-column
// tile), contraction staged through LDS in chunks of 64 with the next chunk's global loads in flight under the current
// chunk's MFMAs (double buffer, ONE barrier per chunk).
//   forward  i = r, j = n, kk = k :  P = x  [i][kk],  Q = w  [j][kk]
//   dX       i = r, j = k, kk = n :  P = dy [i][kk],  Q = w  [kk][j]
//   dW       i = n, j = k, kk = r :  P = dy [kk][i],  Q = x  [kk][j]      (+ db[i] = sum_kk P: fp32, from the staged values)
// An operand whose tile is [outer][kk] is read as a row operand (one 16-byte LDS read per four k-steps), one whose tile
// is [kk][outer] as a gather down a column (four scalar reads): no tile is transposed in memory.
// T = float: k-ordered exact fp32 MFMA (deterministic, the same contraction order for every element); T = bf16_t (bf16
// storage legs, layers.set_storage_dtype): the fp32 operands are rounded to bf16 when they are staged, fp32 accumulate.
constexpr int kLtI = 64, kLtJ = 32, kLtK = 64;

template <class T>
__host__ __device__ inline int lin_tiled_lds_bytes() {
  // P tile 64 x 64 (+ pad), Q tile 32 x 64 or 64 x 32 (+ pad), two buffers
  return 2 * (int)sizeof(T) * (64 * (64 + Lp<T>::PAD) + 64 * (64 + Lp<T>::PAD));
}

struct LinTiledArgs {
  const float* p;   // P source matrix
  const float* q;   // Q source matrix
  const float* bias;
  float* d;         // [I][J]
  float* rowsum;    // db (dW role), nullable
  int ldp, ldq, ldd;
  int I, J, KK;
};

// stage a [ROWS][COLS] fp32 block of a row-major matrix (leading dimension ld) into registers / into an LDS tile of T
template <int ROWS, int COLS>
struct LtStage {
  static constexpr int NV = ROWS * COLS / 4 / kLinThreads;   // float4 per thread
  float4 v[NV];
  __device__ __forceinline__ void load(const float* src, int64_t ld, int r0, int c0) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = threadIdx.x + u * kLinThreads, rr = idx / (COLS / 4), c4 = idx % (COLS / 4);
      v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(r0 + rr) * ld + c0 + 4 * c4);
    }
  }
  template <class T>
  __device__ __forceinline__ void store(T* tile, int pitch) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = threadIdx.x + u * kLinThreads, rr = idx / (COLS / 4), c4 = idx % (COLS / 4);
      Lp<T>::st4(tile + rr * pitch + 4 * c4, v[u].x, v[u].y, v[u].z, v[u].w);
    }
  }
};

// PKK: P tile is [i][kk] (kk contiguous in memory) else [kk][i];  QKK: Q tile is [j][kk] else [kk][j]
template <class T, bool PKK, bool QKK, bool ROWSUM>
__device__ __forceinline__ void lin_tiled_body(const LinTiledArgs& a, int ti, int tj) {
  typedef Lp<T> L;
  typedef typename L::Op Op;
  constexpr int PP = 64 + L::PAD;                       // pitch of the P tile (64 x 64 either way)
  constexpr int QP = (QKK ? kLtK : kLtJ) + L::PAD;      // Q tile: [32 j][64 kk] or [64 kk][32 j]
  constexpr int QROWS = QKK ? kLtJ : kLtK, QCOLS = QKK ? kLtK : kLtJ;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lq = lane & 15, g = lane >> 4;
  const int ih = wv & 1, jt = wv >> 1;
  // (tile addresses are computed from the buffer index, never read from an array of pointers: a pointer loaded from a
  // dynamically indexed array loses its LDS address space, the tile reads become flat loads, and a flat load makes the
  // compiler wait for vmcnt(0) - i.e. for the next chunk's global loads - in front of every MFMA group)
  T* base = reinterpret_cast<T*>(lds_bytes());
  constexpr int PSZ = 64 * PP, QSZ = 64 * (64 + L::PAD);
  const int i0 = kLtI * ti, j0 = kLtJ * tj;
  // Two register sets of staged chunks in flight (A: chunk c + 1, B: chunk c + 2) in front of the LDS double buffer: at
  // one workgroup per CU a single chunk in flight (24 KB per CU) left the kernel bound by the L2 round trip, ~1 us per
  // chunk against 0.45 us of fp32 MFMAs - and with bf16 MFMAs there is nothing else to hide it under.
  LtStage<64, 64> psA, psB;
  LtStage<QROWS, QCOLS> qsA, qsB;
  // (locals, not `a`, inside the lambdas: a lambda that captures a kernel-argument struct by reference can make the
  // compiler keep a copy of the struct in private memory - ffn_bwd.hip)
  const float* gp = a.p;
  const float* gq = a.q;
  const int ldp = a.ldp, ldq = a.ldq;
  auto request = [gp, gq, ldp, ldq, i0, j0](LtStage<64, 64>& ps, LtStage<QROWS, QCOLS>& qs, int c) {
    const int k0 = kLtK * c;
    if (PKK) ps.load(gp, ldp, i0, k0);
    else ps.load(gp, ldp, k0, i0);
    if (QKK) qs.load(gq, ldq, j0, k0);
    else qs.load(gq, ldq, k0, j0);
  };
  // fp32 row sums of P over kk (db of the dW role), from the staged values before they are rounded: this thread's
  // float4 covers four consecutive i of one kk row
  float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto commit = [&rs, base](const LtStage<64, 64>& ps, const LtStage<QROWS, QCOLS>& qs, int buf) {
    ps.template store<T>(base + buf * PSZ, PP);
    qs.template store<T>(base + 2 * PSZ + buf * QSZ, QP);
    if (ROWSUM) {
#pragma unroll
      for (int u = 0; u < LtStage<64, 64>::NV; ++u) {
        rs[0] += ps.v[u].x; rs[1] += ps.v[u].y; rs[2] += ps.v[u].z; rs[3] += ps.v[u].w;
      }
    }
  };
  f32x4 acc[2] = {zero4(), zero4()};
  auto compute = [&acc, base, ih, jt, lq, g](int buf) {
    const T* pt = base + buf * PSZ;
    const T* qt = base + 2 * PSZ + buf * QSZ;
#pragma unroll
    for (int q4 = 0; q4 < kLtK / 16; ++q4) {
      const Op qo = QKK ? L::ld(qt + (16 * jt + lq) * QP + 16 * q4 + 4 * g)
                        : L::gather(qt + (16 * q4 + 4 * g) * QP + 16 * jt + lq, QP);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int il = 32 * ih + 16 * t + lq;
        const Op po = PKK ? L::ld(pt + il * PP + 16 * q4 + 4 * g) : L::gather(pt + (16 * q4 + 4 * g) * PP + il, PP);
        acc[t] = L::mma(po, qo, acc[t]);
      }
    }
  };
  const int nc = a.KK / kLtK;
  request(psA, qsA, 0);
  commit(psA, qsA, 0);
  if (1 < nc) request(psA, qsA, 1);
  if (2 < nc) request(psB, qsB, 2);
  __syncthreads();
  for (int c = 0; c < nc; c += 2) {
    // LDS buffer 0 holds chunk c; A: chunk c + 1, B: chunk c + 2 (in flight)
    compute(0);
    if (c + 1 < nc) commit(psA, qsA, 1);
    if (c + 3 < nc) request(psA, qsA, c + 3);
    __syncthreads();
    if (c + 1 < nc) {
      compute(1);
      if (c + 2 < nc) commit(psB, qsB, 0);
      if (c + 4 < nc) request(psB, qsB, c + 4);
      __syncthreads();
    }
  }
  // D[i][j] (+ bias[j])
  const int col = j0 + 16 * jt + lq;
  const float bv = a.bias != nullptr ? a.bias[col] : 0.0f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) a.d[(int64_t)(i0 + 32 * ih + 16 * t + 4 * g + r) * a.ldd + col] = acc[t][r] + bv;
  if (ROWSUM && a.rowsum != nullptr && tj == 0) {
    // thread (rr = idx / 16, c4 = idx % 16) of vector u summed rows kk = rr + 16 u' ... of columns i0 + 4 c4 ..: the 16
    // threads x NV vectors that share c4 meet in LDS (fixed order)
    float* red = reinterpret_cast<float*>(lds_bytes());     // [256][4] (the tiles are no longer needed)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = rs[e];
    __syncthreads();
    if (threadIdx.x < kLtI) {
      const int c4 = threadIdx.x >> 2, e = threadIdx.x & 3;
      float sv = 0.0f;
      for (int k = 0; k < kLinThreads / 16; ++k) sv += red[(c4 + 16 * k) * 4 + e];
      a.rowsum[i0 + threadIdx.x] = sv;
    }
  }
}

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  The 8 consecutive
// workgroups of a group take 8 different j tiles of ONE i tile, the next group the next 8 j tiles of the same i tile,
// and only then the next i tile: an XCD sees the P rows of an i tile TJ / 8 times in a row (they stay in its L2) and
// keeps cycling over its own TJ / 8 j tiles (its slice of Q).  (Round 3, R = 65536: with the i tiles innermost the
// four users of a P tile on an XCD were 1024 groups apart and the counters showed 8.9 GB for 0.54 GB algorithmic.)
__device__ __forceinline__ void lin_tile_of(int blk, int TI, int TJ, int& ti, int& tj) {
  // blk = (ti * (TJ / 8) + tj_hi) * 8 + tj_lo with tj = tj_hi * 8 + tj_lo  (TJ a multiple of 8)
  const int lo = blk & 7, rest = blk >> 3, nhi = TJ >> 3;
  ti = rest / nhi;
  tj = (rest - ti * nhi) * 8 + lo;
  (void)TI;
}

template <class T>
__global__ __launch_bounds__(kLinThreads) void lin_fwd_tiled_kernel(LinTiledArgs a) {
  int ti, tj;
  lin_tile_of((int)blockIdx.x, a.I / kLtI, a.J / kLtJ, ti, tj);
  lin_tiled_body<T, true, true, false>(a, ti, tj);
}

struct LinBwdTiledArgs {
  LinTiledArgs dx, dw;
  int nx, nw;
  ColsumPlan segs;
};

template <class T>
__global__ __launch_bounds__(kLinThreads) void lin_bwd_tiled_kernel(LinBwdTiledArgs a) {
  int blk = (int)blockIdx.x, ti, tj;
  // the two gradient products interleaved in groups of 8 workgroups, so that both kinds are resident on every CU
  // from the start (two workgroups fit a CU: the dW role's gathers and the dX role's row reads share its LDS pipe)
  const int pairs = a.nx < a.nw ? a.nx : a.nw;
  if (blk < 2 * pairs) {
    const int grp = blk >> 4, in = blk & 15;
    const int sub = grp * 8 + (in & 7);
    if (in < 8) {
      lin_tile_of(sub, a.dx.I / kLtI, a.dx.J / kLtJ, ti, tj);
      lin_tiled_body<T, true, false, false>(a.dx, ti, tj);
    } else {
      lin_tile_of(sub, a.dw.I / kLtI, a.dw.J / kLtJ, ti, tj);
      lin_tiled_body<T, false, false, true>(a.dw, ti, tj);
    }
    return;
  }
  blk -= 2 * pairs;
  if (blk < a.nx - pairs) {
    lin_tile_of(pairs + blk, a.dx.I / kLtI, a.dx.J / kLtJ, ti, tj);
    lin_tiled_body<T, true, false, false>(a.dx, ti, tj);
    return;
  }
  blk -= a.nx - pairs;
  if (blk < a.nw - pairs) {
    lin_tile_of(pairs + blk, a.dw.I / kLtI, a.dw.J / kLtJ, ti, tj);
    lin_tiled_body<T, false, false, true>(a.dw, ti, tj);
    return;
  }
  colsum_role<kLinThreads>(a.segs, blk - (a.nw - pairs));
}

inline bool lin_tiled_ok(int R, int K, int N) {
  // every dim a multiple of 64 and the tile counts over the j dims multiples of 8 (lin_tile_of); FETA_LIN_TILED=0: the
  // one-tile-per-wave form everywhere (A/B timing)
  int64_t min_macs = (int64_t)1 << 24;
  if (const char* e = getenv("FETA_LIN_TILED")) {
    if (atoi(e) == 0) return false;
    if (atoi(e) == 2) min_macs = 0;     // tests: the tiled kernels at the smallest shape they take
  }
  return R % 64 == 0 && K % 256 == 0 && N % 256 == 0 && (int64_t)R * K * N >= min_macs;
}

template <class T>
int launch_lin_fwd_tiled(const float* x, const float* w, const float* bias, float* y, int R, int K, int N,
                         hipStream_t stream) {
  LinTiledArgs a{x, w, bias, y, nullptr, K, K, N, R, N, K};
  auto kern = lin_fwd_tiled_kernel<T>;
  const size_t lds = lin_tiled_lds_bytes<T>();
  static LdsSeen seen;
  allow_dynamic_lds(kern, lds, seen);
  hipLaunchKernelGGL(kern, dim3((R / kLtI) * (N / kLtJ)), dim3(kLinThreads), lds, stream, a);
  return check_launch("feta_lin_fwd");
}

template <class T>
int launch_lin_bwd_tiled(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int R, int K,
                         int N, const ColsumPlan& plan, int tiles, hipStream_t stream) {
  LinBwdTiledArgs a{};
  a.dx = LinTiledArgs{dy, w, nullptr, dx, nullptr, N, K, K, R, K, N};     // i = r, j = k, kk = n
  a.dw = LinTiledArgs{dy, x, nullptr, dw, db, N, K, K, N, K, R};          // i = n, j = k, kk = r
  a.nx = dx != nullptr ? (R / kLtI) * (K / kLtJ) : 0;
  a.nw = (N / kLtI) * (K / kLtJ);
  a.segs = plan;
  auto kern = lin_bwd_tiled_kernel<T>;
  size_t lds = lin_tiled_lds_bytes<T>();
  if (lds < kLinThreads * sizeof(float) * 4) lds = kLinThreads * sizeof(float) * 4;
  static LdsSeen seen;
  allow_dynamic_lds(kern, lds, seen);
  hipLaunchKernelGGL(kern, dim3(a.nx + a.nw + tiles), dim3(kLinThreads), lds, stream, a);
  return check_launch("feta_lin_bwd");
}

}  // namespace feta

using namespace feta;

extern "C" int feta_lin_supported(int R, int K, int N) {
  return (R >= 1 && K >= 16 && N >= 16 && (K & 3) == 0 && (N & 3) == 0 && (R & 3) == 0) ? 1 : 0;
}

extern "C" int feta_lin_fwd(const float* x, const float* w, const float* bias, float* y, int R, int K, int N,
                            feta_stream_t stream) {
  return feta_lin_fwd_ex(x, w, bias, y, R, K, N, FETA_F32, stream);
}

extern "C" int feta_lin_fwd_ex(const float* x, const float* w, const float* bias, float* y, int R, int K, int N,
                               int compute, feta_stream_t stream) {
  FETA_REQUIRE(x && w && y, "lin_fwd: null pointer");
  FETA_REQUIRE(feta_lin_supported(R, K, N), "lin_fwd: need R, K, N multiples of 4, K, N >= 16 (R=%d K=%d N=%d)", R, K, N);
  FETA_REQUIRE(aligned16(x) && aligned16(w), "lin_fwd: x and w must be 16-byte aligned");
  FETA_REQUIRE(compute == FETA_F32 || compute == FETA_BF16, "lin_fwd: compute type %d", compute);
  if (lin_tiled_ok(R, K, N)) {
    if (compute == FETA_BF16) return launch_lin_fwd_tiled<bf16_t>(x, w, bias, y, R, K, N, (hipStream_t)stream);
    return launch_lin_fwd_tiled<float>(x, w, bias, y, R, K, N, (hipStream_t)stream);
  }
  LinFwdArgs a{x, w, bias, y, R, K, N, (N + 31) / 32};
  const int grid = ((R + 31) / 32) * a.tn;
  auto kern = lin_fwd_kernel;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kLinThreads), 0, (hipStream_t)stream, a);
  return check_launch("feta_lin_fwd");
}

extern "C" int feta_lin_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int R,
                            int K, int N, const feta_colsum_seg* segs, int nseg, feta_stream_t stream) {
  return feta_lin_bwd_ex(x, w, dy, dx, dw, db, R, K, N, segs, nseg, FETA_F32, stream);
}

extern "C" int feta_lin_bwd_ex(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int R,
                               int K, int N, const feta_colsum_seg* segs, int nseg, int compute, feta_stream_t stream) {
  FETA_REQUIRE(x && w && dy && dw, "lin_bwd: null pointer");
  FETA_REQUIRE(feta_lin_supported(R, K, N), "lin_bwd: need R, K, N multiples of 4, K, N >= 16 (R=%d K=%d N=%d)", R, K, N);
  FETA_REQUIRE(aligned16(dy), "lin_bwd: dy must be 16-byte aligned");
  FETA_REQUIRE(nseg >= 0 && nseg <= FETA_COLSUM_MAX_SEGS && (nseg == 0 || segs != nullptr),
               "lin_bwd: 0..%d column-sum segments", FETA_COLSUM_MAX_SEGS);
  FETA_REQUIRE(compute == FETA_F32 || compute == FETA_BF16, "lin_bwd: compute type %d", compute);
  for (int i = 0; i < nseg; ++i) FETA_REQUIRE(colsum_seg_ok(segs[i]), "lin_bwd: bad segment %d", i);
  if (lin_tiled_ok(R, K, N)) {
    FETA_REQUIRE(aligned16(x) && aligned16(w), "lin_bwd: x and w must be 16-byte aligned");
    ColsumPlan plan{};
    const int tiles = plan_colsum(segs, nseg, plan);
    if (compute == FETA_BF16)
      return launch_lin_bwd_tiled<bf16_t>(x, w, dy, dx, dw, db, R, K, N, plan, tiles, (hipStream_t)stream);
    return launch_lin_bwd_tiled<float>(x, w, dy, dx, dw, db, R, K, N, plan, tiles, (hipStream_t)stream);
  }
  LinBwdArgs a{};
  a.x = x; a.w = w; a.dy = dy; a.dx = dx; a.dw = dw; a.db = db;
  a.R = R; a.K = K; a.N = N;
  a.tk = (K + 31) / 32;
  a.nx = dx != nullptr ? ((R + 31) / 32) * a.tk : 0;
  a.nw = ((N + 31) / 32) * a.tk;
  const int tiles = plan_colsum(segs, nseg, a.segs);
  auto kern = lin_bwd_kernel;
  hipLaunchKernelGGL(kern, dim3(a.nx + a.nw + tiles), dim3(kLinThreads), kLinThreads * sizeof(float),
                     (hipStream_t)stream, a);
  return check_launch("feta_lin_bwd");
}
