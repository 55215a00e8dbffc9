// Device bodies of the coefficient generator's kernels (coeff.hip) - callable as the kernels themselves or as a
// ROLE in trailing workgroups of another launch: the generator's forward depends only on the attention matrix of the
// last layer, so it can share a launch with that layer's feed-forward half (feta_ffn_fwd_coeff), and its backward,
// which depends only on the filter stage, with the first kernel of the stack's backward (feta_ffn_bwd_coeff): two
// launches per step less, and two half-empty launches become one full one.
#pragma once
#include <cmath>

#include "feta_abi_common.h"
#include <feta_device.h>

namespace feta {

constexpr int kCoeffThreads = 256;
constexpr int kCoeffWideThreads = 1024;   // stand-alone forward launch of graphs beyond 64 nodes: one workgroup per block
constexpr int kCoeffGroupsMax = 128;
constexpr int kCoeffLdsTileMax = 12 * 1024;  // floats of attention staged per block (48 KB)

// body of one (head, graph) block `blk` = h * B + b; TH threads (256 as a kernel of its own for N <= 64 and as a role,
// kCoeffWideThreads for the stand-alone launch of larger graphs), dynamic LDS coeff_fwd_lds_floats_mode(N, stage, TH)
template <int TH = 256>
__device__ __forceinline__ void coeff_fwd_body(
    const float* __restrict__ attn, const int32_t* __restrict__ n_real, const float* __restrict__ s,
    const float* __restrict__ gbias, float* __restrict__ cj_out, float* __restrict__ pooled, int B,
    int N, int H, int C, int stage, int blk, int cs = 0, int CS = 1) {
  // (cs, CS): this workgroup pools channel slice cs of CS - large graphs (N > 64: 128 tanh per channel and block, C = 1024
  // channels) run CS workgroups per block, each recomputing the cheap c_j part; slice 0 writes cj_out
  float* dis = feta_lds;            // [N]
  float* cjs = feta_lds + N;        // [N]
  float* ps = feta_lds + 2 * N;     // [4][64] column partial sums (staged path)
  float* tile = ps + 4 * 64;        // [n][N] when staged
  // blk = h * B + b  (transformer/models.py:244,275,285)
  const int h = blk / B, b = blk % B;
  const int n = n_real[b];
  const float* a = attn + ((int64_t)b * H + h) * N * N;
  const int j = threadIdx.x;

  if (stage && N <= 64) {
    // the block's attention rows in ONE batch of requests (16 per thread cover 64 x 64), then
    // both column sweeps with all 256 threads: thread (column j, slice sl) takes rows i = sl mod 4
    const int cnt = n * N;
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = threadIdx.x + u * TH;
      v[u] = a[idx < cnt ? idx : (cnt > 0 ? cnt - 1 : 0)];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = threadIdx.x + u * TH;
      if (idx < cnt) tile[idx] = v[u];
    }
    __syncthreads();
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    // edges with attn == 0 are dropped (models.py:276,281): they add nothing to the sums,
    // but a dropped self loop is re-created with weight 1 by add_remaining_self_loops.
    float wjj = 0.0f;
    if (col < n) {
      wjj = tile[col * N + col];
      if (wjj == 0.0f) wjj = 1.0f;
    }
    float part = 0.0f;
    if (col < n)
      for (int i = sl; i < n; i += 4) part += (i == col) ? wjj : tile[i * N + col];
    ps[sl * 64 + col] = part;
    __syncthreads();
    if (sl == 0 && col < n) {
      const float deg = (ps[col] + ps[64 + col]) + (ps[128 + col] + ps[192 + col]);
      dis[col] = deg > 0.0f ? rsqrtf(deg) : 0.0f;
    }
    __syncthreads();
    part = 0.0f;
    if (col < n)
      for (int i = sl; i < n; i += 4) part += dis[i] * ((i == col) ? wjj : tile[i * N + col]);
    __syncthreads();   // ps is reused
    ps[sl * 64 + col] = part;
    __syncthreads();
    if (sl == 0 && col < N) {
      float c = 0.0f;
      if (col < n) c = ((ps[col] + ps[64 + col]) + (ps[128 + col] + ps[192 + col])) * dis[col];
      cjs[col] = c;
      if (cs == 0) cj_out[(int64_t)blk * N + col] = c;
    }
    __syncthreads();
  } else if (stage == 2) {
    // graphs beyond 64 nodes (stand-alone launch; config 4): the two column sweeps as float4 column groups x row slices
    // on all 256 threads over the staged rows (pitch NP = N rounded up to 4).  The diagonal rule (a zero self loop
    // counts 1: models.py:276,281 + add_remaining_self_loops) is a correction of the plain column sums:
    //   deg_j = sum_i a_ij + [a_jj == 0],   c_j = dis_j (sum_i dis_i a_ij + [a_jj == 0] dis_j)
    const int NP = (N + 3) & ~3, NC4 = NP >> 2, SL = TH / NC4;
    float* ps2 = feta_lds + 2 * N;     // [SL][NP] (<= 4 TH floats)
    float* tl = ps2 + 4 * TH;          // [n][NP]
    if ((N & 3) == 0) {
      const int cnt4 = n * NC4;
      const float4* a4 = reinterpret_cast<const float4*>(a);
      for (int base = threadIdx.x; base < cnt4; base += 8 * TH) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * TH;
          v[u] = a4[idx < cnt4 ? idx : cnt4 - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * TH;
          if (idx < cnt4) reinterpret_cast<float4*>(tl)[idx] = v[u];   // (NP == N: same linear index)
        }
      }
    } else {
      const int cnt = n * N;
      for (int base = threadIdx.x; base < cnt; base += 8 * TH) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * TH;
          v[u] = a[idx < cnt ? idx : cnt - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * TH;
          if (idx < cnt) tl[(idx / N) * NP + idx % N] = v[u];
        }
      }
      for (int i = threadIdx.x; i < n * (NP - N); i += TH) tl[(i / (NP - N)) * NP + N + i % (NP - N)] = 0.0f;
    }
    __syncthreads();
    const int cg = threadIdx.x % NC4, sl = threadIdx.x / NC4;
    const bool act = sl < SL;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (act)
      for (int i = sl; i < n; i += SL) {
        const float4 v = *reinterpret_cast<const float4*>(tl + i * NP + 4 * cg);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    if (act) *reinterpret_cast<float4*>(ps2 + sl * NP + 4 * cg) = acc;
    __syncthreads();
    float dj = 0.0f, diag0 = 0.0f;
    if (j < n) {
      float deg = 0.0f;
      for (int q = 0; q < SL; ++q) deg += ps2[q * NP + j];
      diag0 = tl[j * NP + j] == 0.0f ? 1.0f : 0.0f;
      deg += diag0;
      dj = deg > 0.0f ? rsqrtf(deg) : 0.0f;
      dis[j] = dj;
    }
    __syncthreads();
    acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (act)
      for (int i = sl; i < n; i += SL) {
        const float4 v = *reinterpret_cast<const float4*>(tl + i * NP + 4 * cg);
        const float di = dis[i];
        acc.x += di * v.x; acc.y += di * v.y; acc.z += di * v.z; acc.w += di * v.w;
      }
    if (act) *reinterpret_cast<float4*>(ps2 + sl * NP + 4 * cg) = acc;
    __syncthreads();
    if (j < N) {
      float c = 0.0f;
      if (j < n) {
        for (int q = 0; q < SL; ++q) c += ps2[q * NP + j];
        c = (c + diag0 * dj) * dj;
      }
      cjs[j] = c;
      if (cs == 0) cj_out[(int64_t)blk * N + j] = c;
    }
    __syncthreads();
  } else {
    const float* src = a;
    if (stage) {
      // eight requests in flight per thread (a plain copy loop pays one memory latency per 256 floats)
      const int cnt = n * N;
      for (int base = threadIdx.x; base < cnt; base += 8 * TH) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * TH;
          v[u] = a[idx < cnt ? idx : cnt - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * TH;
          if (idx < cnt) tile[idx] = v[u];
        }
      }
      __syncthreads();
      src = tile;
    }
    float wjj = 0.0f, deg = 0.0f;
    if (j < n) {
      wjj = src[j * N + j];
      if (wjj == 0.0f) wjj = 1.0f;
      for (int i = 0; i < n; ++i) deg += (i == j) ? wjj : src[i * N + j];
      dis[j] = deg > 0.0f ? rsqrtf(deg) : 0.0f;
    }
    __syncthreads();
    if (j < N) {
      float c = 0.0f;
      if (j < n) {
        for (int i = 0; i < n; ++i) c += dis[i] * ((i == j) ? wjj : src[i * N + j]);
        c *= dis[j];
      }
      cjs[j] = c;
      if (cs == 0) cj_out[(int64_t)blk * N + j] = c;
    }
    __syncthreads();
  }
  const float inv_n = 1.0f / (float)n;
  const int cw = ((C + CS - 1) / CS + TH - 1) / TH * TH;   // channels per slice
  const int cend = min(C, (cs + 1) * cw);
  // four channels of a thread advance together through the node loop: one tanh is a dependent chain of ~6 instructions
  // (two of them transcendental) behind an LDS read - with one chain per iteration and two waves per SIMD (as a role the
  // workgroup inherits its host kernel's 200 registers) the loop ran at the chain's LATENCY, not at the VALU's rate
  if (cw < 4 * TH) {
    // channel slices of large graphs (CS > 1, one channel per thread, thousands of workgroups): throughput-bound on the
    // tanh, so the plain loop - the masked four-wide forms measured 23.5 -> 26.1 us on config 4
    for (int c = cs * cw + threadIdx.x; c < cend; c += TH) {
      const float sc = s[c], bc = gbias[c];
      float acc = 0.0f;
      for (int i = 0; i < n; ++i) acc += fast_tanh(cjs[i] * sc + bc);
      pooled[(int64_t)blk * C + c] = acc * inv_n;
    }
    return;
  }
  for (int c0 = cs * cw + threadIdx.x; c0 < cend; c0 += 4 * TH) {
    float sc[4], bc[4], acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = min(c0 + k * TH, C - 1);
      sc[k] = s[c];
      bc[k] = gbias[c];
      acc[k] = 0.0f;
    }
#pragma unroll 2
    for (int i = 0; i < n; ++i) {
      const float ci = cjs[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += fast_tanh(ci * sc[k] + bc[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * TH;
      if (c < cend) pooled[(int64_t)blk * C + c] = acc[k] * inv_n;
    }
  }
}

// partial[0][grp][c] = sum over the group's blocks of dpooled*(1-z^2)*c_j/n ; partial[1] without c_j.
// A workgroup walks its blocks in passes of kCoeffPass: the dpooled values and c_j rows of a pass are
// requested together (one memory latency per pass, not per block).
constexpr int kCoeffPass = 4;
// body of one (channel tile `bx`, block group `grp`); 256 threads, dynamic LDS kCoeffPass * N floats
__device__ __forceinline__ void coeff_bwd_body(
    const float* __restrict__ cj, const int32_t* __restrict__ n_real, const float* __restrict__ s,
    const float* __restrict__ gbias, const float* __restrict__ dpooled, float* __restrict__ partial,
    int B, int N, int H, int C, int G, int bx, int grp) {
  float* cjs = feta_lds;  // [kCoeffPass][N] c_j rows of the current pass
  const int c = bx * kCoeffThreads + threadIdx.x;
  const int cc = c < C ? c : C - 1;
  const int total = B * H;
  const float sc = s[cc], bc = gbias[cc];
  float as = 0.0f, ab = 0.0f;
  for (int blk0 = grp; blk0 < total; blk0 += G * kCoeffPass) {
    float dp[kCoeffPass];
    int nn[kCoeffPass];
#pragma unroll
    for (int u = 0; u < kCoeffPass; ++u) {
      const int blk = blk0 + u * G;
      const int bc_ = blk < total ? blk : total - 1;
      nn[u] = blk < total ? n_real[bc_ % B] : 0;
      dp[u] = dpooled[(int64_t)bc_ * C + cc];
    }
    __syncthreads();   // the previous pass has been consumed
    for (int i = threadIdx.x; i < kCoeffPass * N; i += kCoeffThreads) {
      const int u = i / N, k = i - u * N;
      const int blk = blk0 + u * G;
      cjs[i] = blk < total ? cj[(int64_t)blk * N + k] : 0.0f;
    }
    __syncthreads();
    // independent tanh chains per iteration (see coeff_fwd_body).  Where the launch is latency-bound (few blocks: the
    // BASELINE batch has 512) the blocks of a pass advance TOGETHER through the node loop - four chains, a block shorter
    // than the longest contributes zeros behind its last node (0.2655 -> 0.262 ms per step); where it is throughput-bound
    // (config 5: 4096 blocks of <= 64 nodes; config 4: 512 blocks of 44..188 nodes) those zeros cost up to twice the tanh
    // count (1.052 -> 1.060 ms on config 5, 32.4 -> 43.1 us for this role on config 4); a masked four-NODE form of one
    // block measured 42.2 us there, so that regime keeps the plain loop.  The switch is the padded tanh count per channel,
    // blocks x N (config 4 sits exactly AT 32768 = 256 x 128, hence the N bound as well).
    int nmax = 0;
#pragma unroll
    for (int u = 0; u < kCoeffPass; ++u) nmax = max(nmax, nn[u]);
    if (N <= 64 && (int64_t)total * N <= 32768) {
      float dpn[kCoeffPass], as_[kCoeffPass], ab_[kCoeffPass];
#pragma unroll
      for (int u = 0; u < kCoeffPass; ++u) {
        dpn[u] = nn[u] > 0 ? dp[u] / (float)nn[u] : 0.0f;
        as_[u] = ab_[u] = 0.0f;
      }
      for (int i = 0; i < nmax; ++i) {
#pragma unroll
        for (int u = 0; u < kCoeffPass; ++u) {
          const float ci = cjs[u * N + i];
          const float z = fast_tanh(ci * sc + bc);
          const float t = i < nn[u] ? dpn[u] * (1.0f - z * z) : 0.0f;
          as_[u] += t * ci;
          ab_[u] += t;
        }
      }
#pragma unroll
      for (int u = 0; u < kCoeffPass; ++u) {
        as += as_[u];
        ab += ab_[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < kCoeffPass; ++u) {
        const int n = nn[u];
        if (n == 0) continue;
        const float dpn = dp[u] / (float)n;
        for (int i = 0; i < n; ++i) {
          const float ci = cjs[u * N + i];
          const float z = fast_tanh(ci * sc + bc);
          const float t = dpn * (1.0f - z * z);
          as += t * ci;
          ab += t;
        }
      }
    }
  }
  if (c < C) {
    partial[(int64_t)grp * 2 * C + c] = as;      // [G][2][C]: one colsum reduces both
    partial[(int64_t)grp * 2 * C + C + c] = ab;
  }
}


// staging mode of a block's attention rows as a ROLE (48 KB tile budget): 1 = N <= 64 (64 x 4 thread layout), 2 = larger
// graphs (float4 column groups x row slices, pitch N rounded up to 4), 0 = not staged.  The stand-alone launch
// (feta_coeff_fwd) stages up to a CU's LDS with the same arithmetic per mode.
__host__ __device__ inline int coeff_fwd_stage(int N) {
  if (N <= 64) return 1;
  return N * ((N + 3) & ~3) <= kCoeffLdsTileMax ? 2 : 0;
}
__host__ __device__ inline int coeff_fwd_lds_floats_mode(int N, int stage, int threads = kCoeffThreads) {
  return 2 * N + (stage == 2 ? 4 * threads + N * ((N + 3) & ~3) : 4 * 64 + (stage == 1 ? N * N : 0));
}
__host__ __device__ inline int coeff_fwd_lds_floats(int N) { return coeff_fwd_lds_floats_mode(N, coeff_fwd_stage(N)); }

// plain argument blocks of the two roles (device pointers)
struct CoeffFwdRole {
  const float* attn; const int32_t* n_real; const float* s; const float* gbias; float* cj; float* pooled;
  int B, N, H, C;
};
struct CoeffBwdRole {
  const float* cj; const int32_t* n_real; const float* s; const float* gbias; const float* dpooled; float* partial;
  int B, N, H, C, G;
};

}  // namespace feta
