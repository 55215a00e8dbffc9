// Producer side of the spectral path: the per-graph symmetric
// eigendecomposition  A_b = U_b diag(lam_b) U_b^T  and kernel functions of the spectrum
// U f(lam) U^T, batched on the device.  The reference does both on the host, one graph at a time
// (np.linalg.eig per graph, transformer/position_encoding.py:127-161; scipy expm / sparse matrix
// powers per graph, :65-72 and :83-93) and caches the result in a pickle (:11-52).
//
// feta_eigh_sym: one workgroup per graph, the whole matrix in LDS (N <= 192; for 192 < N <= 256 - 266 KB, more than
// a CU's 160 KB - in a caller-provided workspace that stays L2-resident: same code, the matrix pointer is global
// memory and the round barrier orders the workgroup's own stores and loads), one-sided (Hestenes) Jacobi.
// G = A + shift*I is symmetric positive definite (the caller's contract), so the rotations that
// orthogonalise the columns of G from the right, G <- G J, end at G = (A + shift) V with
// V^T (A+shift)^2 V diagonal: the columns are  g_j = sigma_j v_j,  sigma_j = lam_j + shift > 0, and
// the eigenvectors are the normalised columns themselves - no second matrix is accumulated.
// Column pairs of one round of a round-robin tournament are disjoint, so a round is one barrier;
// a pair is owned by 16 lanes (one DPP row: the three dot products are row sums on the VALU) that
// hold 4 consecutive rows each (one ds_read_b128 per column and 64 rows).
#include "feta_abi_common.h"
#include <feta_device.h>

namespace feta {

struct EighArgs {
  const float* a;         // [B, N, N]
  const int32_t* n_real;  // [B]
  float shift, tol;
  int K;
  float* u;          // [B, N, K]
  float* lam;        // [B, K]
  int32_t* sweeps;   // [B] or null
  float* work;       // [B, N, pitch] matrix storage when it does not fit in LDS, else null
  int B, N, max_sweeps;
};

constexpr int kEighMaxSweeps = 30;

__host__ __device__ inline int eigh_pitch(int NC) { return 64 * NC + 4; }   // floats per column: 4 * odd
inline size_t eigh_lds_floats(int NC, int N) {
  return (size_t)eigh_pitch(NC) * N + 3 * (size_t)N + kEighMaxSweeps + 2;
}

// round `step` of the tournament over m (even) players, table `k` of m/2 -> the two players
__device__ __forceinline__ void tournament_pair(int step, int k, int m, int& p, int& q) {
  if (k == 0) {
    p = m - 1;
    q = step;
  } else {
    p = step + k;             // both in [0, 2 (m - 1)): one conditional subtraction instead of a division
    q = step - k + (m - 1);
    if (p >= m - 1) p -= m - 1;
    if (q >= m - 1) q -= m - 1;
  }
}

template <int NC, bool GLOBAL>
__global__ __launch_bounds__(512) void eigh_jacobi_kernel(EighArgs a) {
  constexpr int P = 64 * NC + 4;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int l = tid & 15, grp = tid >> 4, ngrp = nthr >> 4;
  const int b = blockIdx.x;
  const int n = min(a.n_real[b], a.N);
  // column j at G + j * P, rows [0, 64 NC): in LDS, or (GLOBAL) in this graph's slice of the workspace
  float* G = GLOBAL ? a.work + (size_t)b * P * a.N : feta_lds;
  float* lamv = GLOBAL ? feta_lds : G + (size_t)P * a.N;        // [N] eigenvalue of column j
  float* scl = lamv + a.N;                  // [N] sign / norm of column j
  int* perm = reinterpret_cast<int*>(scl + a.N);   // [N] column holding the k-th smallest eigenvalue
  int* flags = perm + a.N;                  // [kEighMaxSweeps + 1] "a rotation happened in sweep s"

  // ---- G = lower triangle of A mirrored (numpy.linalg.eigh's UPLO='L') + shift on the diagonal ----
  const float* A = a.a + (int64_t)b * a.N * a.N;
  for (int idx = tid; idx < n * (64 * NC); idx += nthr) {
    const int j = idx / (64 * NC), i = idx - j * (64 * NC);
    float v = 0.0f;
    if (i < n) {
      v = i >= j ? A[(int64_t)i * a.N + j] : A[(int64_t)j * a.N + i];
      if (i == j) v += a.shift;
    }
    G[j * P + i] = v;
  }
  for (int s = tid; s <= kEighMaxSweeps; s += nthr) flags[s] = 0;
  __syncthreads();

  const int m = n + (n & 1), half = m >> 1;
  const float tol2 = a.tol * a.tol;
  int sweep = 0;
  for (; sweep < a.max_sweeps; ++sweep) {
    for (int step = 0; step < m - 1; ++step) {
      for (int k0 = 0; k0 < half; k0 += ngrp) {   // (trip count uniform over the wave: DPP rows stay whole)
        const int k = k0 + grp;
        int p = 0, q = 0;
        if (k < half) tournament_pair(step, k, m, p, q);
        const bool live = k < half && p < n && q < n;   // not the bye of an odd n
        if (!live) p = q = 0;                           // reads column 0, writes nothing
        float4 gp[NC], gq[NC];
        float al = 0.0f, be = 0.0f, ga = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          gp[c] = *reinterpret_cast<const float4*>(G + p * P + 64 * c + 4 * l);
          gq[c] = *reinterpret_cast<const float4*>(G + q * P + 64 * c + 4 * l);
          al += gp[c].x * gp[c].x + gp[c].y * gp[c].y + gp[c].z * gp[c].z + gp[c].w * gp[c].w;
          be += gq[c].x * gq[c].x + gq[c].y * gq[c].y + gq[c].z * gq[c].z + gq[c].w * gq[c].w;
          ga += gp[c].x * gq[c].x + gp[c].y * gq[c].y + gp[c].z * gq[c].z + gp[c].w * gq[c].w;
        }
        al = row16_sum(al);
        be = row16_sum(be);
        ga = row16_sum(ga);
        if (live && ga * ga > tol2 * al * be) {   // same value on the 16 lanes of the pair
          // (hardware reciprocal / square root: the angle only steers convergence, and the scale of
          // (c, s) is repaired below)
          const float zeta = 0.5f * (be - al) * fast_rcp(ga);
          const float t = copysignf(fast_rcp(fabsf(zeta) + fast_sqrt(1.0f + zeta * zeta)), zeta);
          const float cs = fast_rsqrt(1.0f + t * t), sn = cs * t;
          // c^2 + s^2 = 1 only to a few ulp, and that scale error is common to every element of both
          // columns: over the ~n * sweeps rotations a column goes through it would drift the norms
          // (= the eigenvalues) by tens of ulp.  The defect e = 1 - c^2 - s^2 is evaluated exactly
          // (fma residuals) and (c, s) carry the first-order correction (1 + e/2) in a low word.
          const float c2 = cs * cs, s2 = sn * sn;
          const float e = ((1.0f - c2) - s2) - fmaf(cs, cs, -c2) - fmaf(sn, sn, -s2);
          const float cl = 0.5f * e * cs, sl = 0.5f * e * sn;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            float4 np_, nq_;
            np_.x = fmaf(cs, gp[c].x, fmaf(-sn, gq[c].x, cl * gp[c].x - sl * gq[c].x));
            np_.y = fmaf(cs, gp[c].y, fmaf(-sn, gq[c].y, cl * gp[c].y - sl * gq[c].y));
            np_.z = fmaf(cs, gp[c].z, fmaf(-sn, gq[c].z, cl * gp[c].z - sl * gq[c].z));
            np_.w = fmaf(cs, gp[c].w, fmaf(-sn, gq[c].w, cl * gp[c].w - sl * gq[c].w));
            nq_.x = fmaf(sn, gp[c].x, fmaf(cs, gq[c].x, sl * gp[c].x + cl * gq[c].x));
            nq_.y = fmaf(sn, gp[c].y, fmaf(cs, gq[c].y, sl * gp[c].y + cl * gq[c].y));
            nq_.z = fmaf(sn, gp[c].z, fmaf(cs, gq[c].z, sl * gp[c].z + cl * gq[c].z));
            nq_.w = fmaf(sn, gp[c].w, fmaf(cs, gq[c].w, sl * gp[c].w + cl * gq[c].w));
            *reinterpret_cast<float4*>(G + p * P + 64 * c + 4 * l) = np_;
            *reinterpret_cast<float4*>(G + q * P + 64 * c + 4 * l) = nq_;
          }
          if (l == 0) flags[sweep] = 1;
        }
      }
      __syncthreads();
    }
    __syncthreads();   // (n <= 1: no round ran) the flag of this sweep is final
    if (flags[sweep] == 0) break;   // same answer on every thread: read after the barrier
  }
  if (a.sweeps != nullptr && tid == 0) a.sweeps[b] = sweep;

  // ---- eigenvalue and normalisation of each column; sign: the entry of largest magnitude
  // (lowest row on ties) is positive ----
  for (int j0 = 0; j0 < n; j0 += ngrp) {
    const int j = min(j0 + grp, n - 1);
    float ss = 0.0f, best = -1.0f, bestv = 0.0f, besti = 0.0f;   // (row index as float: exact, one shuffle type)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float4 g4 = *reinterpret_cast<const float4*>(G + j * P + 64 * c + 4 * l);
      const float e[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ss += e[r] * e[r];
        if (fabsf(e[r]) > best) {
          best = fabsf(e[r]);
          bestv = e[r];
          besti = (float)(64 * c + 4 * l + r);
        }
      }
    }
    ss = row16_sum(ss);
#pragma unroll
    for (int msk = 1; msk < 16; msk <<= 1) {
      const float ob = shfl_xor(best, msk), ov = shfl_xor(bestv, msk), oi = shfl_xor(besti, msk);
      if (ob > best || (ob == best && oi < besti)) {
        best = ob;
        bestv = ov;
        besti = oi;
      }
    }
    if (l == 0 && j0 + grp < n) {
      const float sigma = sqrtf(ss);
      lamv[j] = sigma - a.shift;
      scl[j] = (bestv < 0.0f ? -1.0f : 1.0f) / sigma;
    }
  }
  __syncthreads();
  // ---- ascending order (ties: column index), by counting ----
  for (int j = tid; j < n; j += nthr) {
    const float lj = lamv[j];
    int rank = 0;
    for (int i = 0; i < n; ++i) {
      const float li = lamv[i];
      rank += (li < lj || (li == lj && i < j)) ? 1 : 0;
    }
    perm[rank] = j;
  }
  __syncthreads();
  // ---- u [N, K] (zero rows / columns beyond n), lam [K] ----
  float* U = a.u + (int64_t)b * a.N * a.K;
  for (int idx = tid; idx < a.N * a.K; idx += nthr) {
    const int i = idx / a.K, k = idx - i * a.K;
    float v = 0.0f;
    if (i < n && k < n) {
      const int j = perm[k];
      v = G[j * P + i] * scl[j];
    }
    U[idx] = v;
  }
  for (int k = tid; k < a.K; k += nthr) a.lam[(int64_t)b * a.K + k] = k < n ? lamv[perm[k]] : 0.0f;
}

// ---- out_b = U_b f(lam_b) U_b^T on the real block ------------------------------------------------

struct SpecFnArgs {
  const float* u;         // [B, N, K]
  const float* lam;       // [B, K]
  const int32_t* n_real;  // [B]
  float* out;             // [B, N, N]
  float beta, lam_offset;
  int mode, p, zero_diag;
  int B, N, K;
};

__host__ __device__ inline int specfn_pitch(int K) { return 16 * ((K + 15) / 16) + 4; }   // 4 * odd
inline size_t specfn_lds_floats(int N, int K) {
  return (size_t)specfn_pitch(K) * (16 * ((N + 15) / 16)) + 16 * ((K + 15) / 16);
}

__global__ __launch_bounds__(256) void spectral_fn_kernel(SpecFnArgs a) {
  const int tid = threadIdx.x, lane = lane_id(), lq = lane & 15, g = lane >> 4, wv = wave_id();
  const int b = blockIdx.x;
  const int n = min(a.n_real[b], a.N);
  const int KP = specfn_pitch(a.K), K16 = KP - 4;
  const int NTall = (a.N + 15) >> 4, NT = (n + 15) >> 4;
  float* Us = feta_lds;                    // [16 NTall][KP], zero beyond (n, min(K, n))
  float* fk = Us + (size_t)KP * 16 * NTall;   // [K16]
  const float* U = a.u + (int64_t)b * a.N * a.K;
  for (int idx = tid; idx < 16 * NT * K16; idx += 256) {
    const int i = idx / K16, k = idx - i * K16;
    Us[i * KP + k] = (i < n && k < a.K && k < n) ? U[(int64_t)i * a.K + k] : 0.0f;
  }
  for (int k = tid; k < K16; k += 256) {
    float f = 0.0f;
    if (k < a.K && k < n) {
      const float x = a.lam[(int64_t)b * a.K + k] + a.lam_offset;
      if (a.mode == 0) {
        f = expf(-a.beta * x);          // diffusion: expm(-beta L)
      } else {
        const float base = 1.0f - a.beta * x;   // p-step random walk: (I - beta L)^p
        f = base;   // (the reference multiplies p - 1 times: p = 0 is the first power as well)
        for (int e = 1; e < a.p; ++e) f *= base;
      }
    }
    fk[k] = f;
  }
  __syncthreads();
  float* O = a.out + (int64_t)b * a.N * a.N;
  for (int t = wv; t < NTall * NTall; t += 4) {
    const int it = t / NTall, jt = t - it * NTall;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (it < NT && jt < NT) {
      // instruction r of block kb contracts k = 16 kb + 4 g + r: a permutation of k shared by both
      // operands, so each lane reads 16 bytes per operand and block
      for (int kb = 0; kb < K16 / 16; ++kb) {
        const float4 u4 = *reinterpret_cast<const float4*>(Us + (16 * it + lq) * KP + 16 * kb + 4 * g);
        const float4 v4 = *reinterpret_cast<const float4*>(Us + (16 * jt + lq) * KP + 16 * kb + 4 * g);
        const float4 f4 = *reinterpret_cast<const float4*>(fk + 16 * kb + 4 * g);
        acc = mfma16(u4.x * f4.x, v4.x, acc);
        acc = mfma16(u4.y * f4.y, v4.y, acc);
        acc = mfma16(u4.z * f4.z, v4.z, acc);
        acc = mfma16(u4.w * f4.w, v4.w, acc);
      }
    }
    const int j = 16 * jt + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * it + 4 * g + r;
      if (i < a.N && j < a.N) {
        const bool keep = i < n && j < n && !(a.zero_diag && i == j);
        O[(int64_t)i * a.N + j] = keep ? acc[r] : 0.0f;
      }
    }
  }
}

}  // namespace feta

using namespace feta;

extern "C" int feta_eigh_sym_supported(int N) { return N >= 1 && N <= 256 ? 1 : 0; }

extern "C" int64_t feta_eigh_sym_workspace_bytes(int B, int N) {
  if (N <= 192) return 0;
  return (int64_t)sizeof(float) * B * N * eigh_pitch((N + 63) / 64);
}

extern "C" int feta_eigh_sym(const float* a, const int32_t* n_real, float shift, float* u, float* lam,
                             int32_t* sweeps, float* workspace, int B, int N, int K, int max_sweeps, float tol,
                             feta_stream_t stream) {
  FETA_REQUIRE(a && n_real && u && lam && B > 0 && K >= 1 && K <= N, "eigh_sym: bad arguments");
  FETA_REQUIRE(feta_eigh_sym_supported(N), "eigh_sym: N = %d not supported (1..256)", N);
  FETA_REQUIRE(N <= 192 || (workspace != nullptr && aligned16(workspace)),
               "eigh_sym: N = %d needs a 16-byte aligned workspace of feta_eigh_sym_workspace_bytes(B, N)", N);
  EighArgs args;
  args.work = N > 192 ? workspace : nullptr;
  args.a = a;
  args.n_real = n_real;
  args.shift = shift;
  args.tol = tol > 0.0f ? tol : 1e-6f;
  args.K = K;
  args.u = u;
  args.lam = lam;
  args.sweeps = sweeps;
  args.B = B;
  args.N = N;
  args.max_sweeps = max_sweeps > 0 ? (max_sweeps < kEighMaxSweeps ? max_sweeps : kEighMaxSweeps) : 16;
  const int NC = (N + 63) / 64;
  const size_t lds = N > 192 ? sizeof(float) * (3 * (size_t)N + kEighMaxSweeps + 2)
                             : sizeof(float) * eigh_lds_floats(NC, N);
  const dim3 grid(B), block(N > 32 ? 512 : 256);
  if (N > 192) {
    auto kern = eigh_jacobi_kernel<4, true>;
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, args);
    return check_launch("feta_eigh_sym");
  }
#define FETA_EIGH_CASE(nc)                                                           \
  case nc: {                                                                         \
    auto kern = eigh_jacobi_kernel<nc, false>;                                       \
    static LdsSeen seen;                                                          \
    allow_dynamic_lds(kern, lds, seen);                                              \
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)stream, args);           \
  } break;
  switch (NC) {
    FETA_EIGH_CASE(1)
    FETA_EIGH_CASE(2)
    FETA_EIGH_CASE(3)
  }
#undef FETA_EIGH_CASE
  return check_launch("feta_eigh_sym");
}

extern "C" int feta_spectral_kernel(const float* u, const float* lam, const int32_t* n_real, int mode,
                                    float beta, int p, float lam_offset, int zero_diag, float* out,
                                    int B, int N, int K, feta_stream_t stream) {
  FETA_REQUIRE(u && lam && n_real && out && B > 0 && N >= 1 && K >= 1 && K <= N, "spectral_kernel: bad arguments");
  FETA_REQUIRE(mode == FETA_SPECTRAL_DIFFUSION || mode == FETA_SPECTRAL_PSTEP, "spectral_kernel: unknown mode %d", mode);
  FETA_REQUIRE(mode != FETA_SPECTRAL_PSTEP || p >= 0, "spectral_kernel: p = %d", p);
  const size_t lds = sizeof(float) * specfn_lds_floats(N, K);
  FETA_REQUIRE(lds <= 160 * 1024, "spectral_kernel: N = %d, K = %d does not fit in LDS", N, K);
  SpecFnArgs args;
  args.u = u;
  args.lam = lam;
  args.n_real = n_real;
  args.out = out;
  args.beta = beta;
  args.lam_offset = lam_offset;
  args.mode = mode;
  args.p = p;
  args.zero_diag = zero_diag;
  args.B = B;
  args.N = N;
  args.K = K;
  auto kern = spectral_fn_kernel;
  static LdsSeen seen;
  allow_dynamic_lds(kern, lds, seen);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, args);
  return check_launch("feta_spectral_kernel");
}
