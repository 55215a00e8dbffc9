// A1, the attention sub-block of DiffTransformerEncoderLayer as ONE launch per layer for the
// BASELINE shape (d = 64 = 4 heads x 16, N <= 64): in_proj -> attention core -> out_proj ->
// degree scale -> residual -> BatchNorm statistics (contract transformer/models.py:166-167,179,244;
// body per upstream GraphiT, README.md:129).  One workgroup per graph, one wave per head.
//
// The three stages share everything on chip:
//   * the graph's node rows X_b [N, 64] are fetched once as whole 256-byte rows, normalised on the
//     way in (the BatchNorm of the previous layer is folded into this load; the first consumer of
//     fresh statistics finalizes them), and stay in LDS - they are the in_proj operand AND the
//     residual of the out_proj epilogue;
//   * W_in [192, 64] and W_out [64, 64] are staged once per workgroup (padded pitch: 16-byte operand
//     reads hit disjoint banks for the two 4-row groups of a half-wave);
//   * Q^T / K^T tiles come out of the MFMA in exactly the lane layout the score product takes as
//     "row operands", V in the layout P.V takes as B operand (feta_tiles.h) - no shuffles, no LDS;
//   * the per-head outputs meet in an LDS tile [N, 64] which is the out_proj operand; wave w then
//     owns output columns 16w .. 16w+15, so the column statistics need no cross-wave reduction.
// q, k, v, the per-head output and the softmax statistics are still written to HBM: the backward
// pass (attn.hip, rowwise.hip) and the spectral filter read them.
#include <cstdlib>

#include "feta_abi_common.h"
#include "feta_colsum.h"
#include "feta_ln.h"
#include "feta_lp.h"
#include "feta_rowops.h"

namespace feta {

typedef feta_attn_block BlockArgs;  // include/feta_hip.h

constexpr int kBlkD = 64, kBlkH = 4, kBlkDH = 16;
constexpr int kBlkP = kBlkD + 4;  // LDS pitch of every staged 64-float row
constexpr int kBlkMaxGrid = 256;  // workgroups of a launch (MI355X: 256 CUs, one such workgroup each)

// LDS of a workgroup, in bytes: tiles of T (pitch kBlkD + Lp<T>::PAD elements), fp32 for everything else
template <class T>
__host__ __device__ inline int block_lds_bytes(int nt, bool attn) {
  const int nr = 16 * nt, P = kBlkD + Lp<T>::PAD;
  int b = (int)sizeof(T) * (3 * kBlkD * P + kBlkD * P);   // W_in, W_out
  b += (int)sizeof(T) * 2 * nr * P;                          // X tile, OUT tile
  int f = 2 * kBlkD;                                         // scale / shift of the input BatchNorm
  const int fin = reduce_scratch_floats(kBlkD);
  const int stg = attn ? kBlkH * 16 * (nr + 1) : 0;          // per-wave probability staging
  f += fin > stg ? fin : stg;
  f += nr * (nr + 4);                                        // pe tile (fp32 whatever the storage type)
  return b + 4 * f;
}

#ifdef FETA_TIMING
__device__ unsigned long long feta_block_stamps[4 * 8 * 8];   // FETA_RT_STAMP (feta_rowops.h), tools/block_timing.py
__device__ unsigned int feta_block_launch;
#endif
#define FETA_STAMP(i) FETA_RT_STAMP(feta_block_stamps, feta_block_launch, i)

// Workgroups beyond main_grid reduce the column sums of `sums` (feta_colsum.h): the first launch of a forward pass
// leaves half of the chip idle at the BASELINE batch, and s = colsum(gcn.weight) of the coefficient generator - a
// function of the parameters alone - would otherwise be a launch of its own (~7 us).
// T: storage type of x, pe, qkv, out, y and of the LDS tiles (feta_lp.h); weights, biases, statistics, attn: fp32.
template <class T, int NT>
__global__ __launch_bounds__(kRowThreads) void attn_block_fwd_kernel(BlockArgs a, ColsumPlan sums, int main_grid,
                                                                    int weights_last) {
  typedef Lp<T> L;
  typedef typename L::Op Op;
  typedef typename L::Vec Vec;
  constexpr int D = kBlkD, DH = kBlkDH, P = kBlkD + L::PAD, NR = 16 * NT, KP = NR + 1;
  constexpr int RV = D / L::VEC;                                  // 16-byte vectors of a 64-element row
  constexpr int XI = (NR * RV + kRowThreads - 1) / kRowThreads;   // ... of the graph's rows, per thread
  if ((int)blockIdx.x >= main_grid) {
    colsum_role<kRowThreads>(sums, (int)blockIdx.x - main_grid);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, lq = lane & 15, g = lane >> 4;
  T* Wi = reinterpret_cast<T*>(lds_bytes());   // [192][P]
  T* Wo = Wi + 3 * D * P;                      // [64][P]
  T* Xs = Wo + D * P;                          // [NR][P]  layer input, BatchNorm applied
  T* Os = Xs + NR * P;                         // [NR][P]  per-head outputs (concat)
  float* xss = reinterpret_cast<float*>(Os + NR * P);   // [2][64]
  float* scr = xss + 2 * D;                    // finalize scratch, later the probability staging
  constexpr int PEP = NR + 4;                  // pitch of the pe tile (16-byte operand reads)
  float* Pe = reinterpret_cast<float*>(lds_bytes() + block_lds_bytes<T>(NT, a.attn != nullptr)) - NR * PEP;   // [NR][PEP]
  const T* gx = reinterpret_cast<const T*>(a.x);
  const T* gpe = reinterpret_cast<const T*>(a.pe);
  T* gqkv = reinterpret_cast<T*>(a.qkv);
  T* gout = reinterpret_cast<T*>(a.out);
  T* gy = reinterpret_cast<T*>(a.y);
  const bool x_norm = a.x_stats != nullptr || a.x_bn != nullptr;
  FETA_STAMP(0);

  // ---- once per workgroup: biases, W_in / W_out into LDS, BatchNorm of the input finalized; the
  // workgroup then walks its graphs (b, b + gridDim.x, ...) with the weights in place - at large
  // batches the 64 KB of weights per graph would otherwise be the largest stream of the kernel
  // The global loads of the prologue are requested before the first one is consumed (partial statistics, the first
  // graph's rows, biases, weights: in-order returns): after a kernel boundary each first touch is a ~1-2 us round trip,
  // and they used to run one after the other.
  PartialBatch pb;   // (requested unconditionally - a conditionally filled array lives in scratch; G = 0 reads row 0 of
                     // a tensor that is always there)
  partials_request(a.x_stats != nullptr ? a.x_stats : a.w_in, a.x_stats != nullptr ? a.Gx : 0, D, pb);
  const bool has_pe = a.pe != nullptr;
  constexpr int PEI = (NR * NR + kRowThreads - 1) / kRowThreads;
  Vec xv[XI];
  float pel[PEI];
  float rsv[NT];
  int n = 0;
  const int nn = a.N * a.N;
  // node rows, the graph's pe block as ONE coalesced stream (it is contiguous: N x N elements; every head needs all of
  // it - four waves gathering their (query, key) pairs one element at a time took ~2.5 us, and a __syncthreads waits for
  // every load in flight: vmcnt counts loads and stores alike on gfx9), the degree scale of this lane's rows
  auto request_graph = [&](int b) {
    n = a.n_real[b];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int idx = min(tid + kRowThreads * i, NR * RV - 1), node = idx / RV, q = idx % RV;
      const int64_t row = (int64_t)b * a.row_sb + (int64_t)min(node, a.N - 1) * a.row_sn;
      xv[i] = L::ldv(gx + row * D + L::VEC * q);
    }
#pragma unroll
    for (int i = 0; i < PEI; ++i)
      pel[i] = has_pe ? L::ld1(gpe + (int64_t)b * nn + min(tid + kRowThreads * i, nn - 1)) : 1.0f;
#pragma unroll
    for (int qb = 0; qb < NT; ++qb) {
      const int qc = min(16 * qb + lq, a.N - 1);
      rsv[qb] = a.rowscale != nullptr ? a.rowscale[(int64_t)b * a.row_sb + (int64_t)qc * a.row_sn] : 1.0f;
    }
  };
  request_graph(blockIdx.x);
  float4 bin4[3], bo = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float bin1 = 0.0f;
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    bin4[part] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a.b_in != nullptr) bin4[part] = *reinterpret_cast<const float4*>(a.b_in + part * D + DH * h + 4 * g);
  }
  if (a.b_in != nullptr) bin1 = a.b_in[2 * D + DH * h + lq];
  if (a.b_out != nullptr) bo = *reinterpret_cast<const float4*>(a.b_out + DH * h + 4 * g);
  // 256 rows of 16 float4 (W_in then W_out: fp32 masters), 16 per thread
  float4 wv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = tid + kRowThreads * i, r = idx >> 4, q = idx & 15;
    const float* src = r < 3 * D ? a.w_in + (int64_t)r * D : a.w_out + (int64_t)(r - 3 * D) * D;
    wv[i] = *reinterpret_cast<const float4*>(src + 4 * q);
  }
  float xg = 1.0f, xb = 0.0f, xk = 0.0f;
  if (a.x_stats != nullptr && tid < D) {
    xg = a.x_gamma[tid];
    xb = a.x_beta[tid];
    xk = partials_shift(a.x_stats, a.Gx, D, tid);
  }
  // Loads return in request order: the partial statistics and the first graph's rows are here long before the 64 KB of
  // weights.  Round 3 experiment (FETA_BLOCK_WEIGHTS_LAST=1): they are CONSUMED in that order too - statistics finalized
  // and the first graph staged while the weights are still travelling, the weights stored to LDS last; the barriers in
  // between order LDS traffic only (lds_barrier: a __syncthreads() behind the published-parameter stores of workgroup 0
  // waits for vmcnt(0), i.e. for the weights).  It bought nothing (see the launcher): the default stays weights first.
  auto store_weights = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = tid + kRowThreads * i;
      L::st4(Wi + (idx >> 4) * P + 4 * (idx & 15), wv[i].x, wv[i].y, wv[i].z, wv[i].w);   // (rounded once, here, for bf16)
    }
  };
  if (!weights_last) store_weights();
  FETA_STAMP(6);
  if (a.x_stats != nullptr) {
    // first consumer of fresh statistics: every workgroup finalizes them (redundantly and
    // deterministically); workgroup 0 publishes the parameter block and the running statistics
    reduce_partials_finish(a.x_stats, a.Gx, D, pb, scr + 2 * D, scr);
    for (int c = tid; c < D; c += kRowThreads) {
      float mean, var;
      bn_moments_k(xk, D, a.M, scr, c, mean, var);   // (c == tid: the loop runs once for the first D threads)
      const float rstd = rsqrtf(var + a.eps);
      const float scale = xg * rstd;
      const float shift = xb - mean * scale;
      xss[c] = scale;
      xss[D + c] = shift;
      if (blockIdx.x == 0) {
        a.x_bn_out[c] = scale;
        a.x_bn_out[D + c] = shift;
        a.x_bn_out[2 * D + c] = mean;
        a.x_bn_out[3 * D + c] = rstd;
        if (a.x_rmean != nullptr) {
          const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
          a.x_rmean[c] = (1.0f - a.momentum) * a.x_rmean[c] + a.momentum * mean;
          a.x_rvar[c] = (1.0f - a.momentum) * a.x_rvar[c] + a.momentum * unbiased;
        }
        if (c == 0 && a.x_nbt != nullptr) *a.x_nbt += 1;
      }
    }
  } else if (a.x_bn != nullptr) {
    for (int c = tid; c < 2 * D; c += kRowThreads) xss[c] = a.x_bn[c];
  } else if (a.x_ln_gamma != nullptr) {   // LayerNorm on load (feta_ln.h): xss = gamma | beta
    for (int c = tid; c < 2 * D; c += kRowThreads) xss[c] = c < D ? a.x_ln_gamma[c] : a.x_ln_beta[c - D];
  }
  const bool x_ln = a.x_ln_gamma != nullptr;
  FETA_STAMP(7);
  bool first = true;
  for (int b = blockIdx.x; b < a.B; b += main_grid) {
  if (!first) {
    __syncthreads();   // the tiles of the previous graph have been consumed
    request_graph(b);
  }
  const bool late_weights = first && weights_last;
  first = false;
  if (late_weights) lds_barrier();   // (xss is LDS data; the weights stay in flight)
  else __syncthreads();
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = tid + kRowThreads * i, node = idx / RV, q = idx % RV;
    if (XI * kRowThreads != NR * RV && idx >= NR * RV) continue;
    Vec v = xv[i];
    if (x_norm) {
      float f[L::VEC];
      L::unpack(v, f);
#pragma unroll
      for (int e = 0; e < L::VEC; ++e) f[e] = f[e] * xss[L::VEC * q + e] + xss[D + L::VEC * q + e];
      v = L::pack(f);
    } else if (x_ln) {
      // the row is held by RV consecutive lanes: mean / rstd by DPP sums over them, then gamma / beta (a wave-uniform
      // branch: a wave stages whole rows)
      float f[L::VEC];
      L::unpack(v, f);
      ln_apply<L::VEC>(f, xss + L::VEC * q, xss + D + L::VEC * q, a.eps);
      v = L::pack(f);
    }
    L::stv(Xs + node * P + L::VEC * q, v);  // rows >= N: a copy of row N-1, never stored
  }
  {
    const float rn = 1.0f / (float)a.N;
#pragma unroll
    for (int i = 0; i < PEI; ++i) {
      const int idx = tid + kRowThreads * i;
      // idx / N: (idx + 1/2) / N is at least 1 / (2 N) away from an integer, far beyond the rounding of the product
      const int qq = (int)(((float)idx + 0.5f) * rn), kk = idx - qq * a.N;
      if (idx < nn) Pe[qq * PEP + kk] = pel[i];
    }
  }
  if (late_weights) store_weights();
  __syncthreads();
  FETA_STAMP(1);

  // ---- in_proj for this head: Q^T (scaled), K^T ("row operand" layout), V (B-operand layout) -----
  Op qs[NT], kf[NT], vbo[NT];
  {
    RowOp<T, D> xf[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) load_row_op<T, D>(xf[nt], Xs + (16 * nt + lq) * P, g);
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      if (part == 1 && a.tie_qk) continue;
      RowOp<T, D> wf;
      load_row_op<T, D>(wf, Wi + (part * D + DH * h + lq) * P, g);
      const float4 bv4 = bin4[part];
      const float bv1 = bin1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int node = 16 * nt + lq;
        if (part > 0 && 16 * nt >= n) {  // a key tile without a real node: no K, no V (wave-uniform)
          if (part == 1) kf[nt] = L::zero();
          else vbo[nt] = L::zero();
          continue;
        }
        if (part < 2) {
          // (c = 4g + r, node = lq): four consecutive features of one node per lane
          f32x4 t = dot_row_ops<T, D>(wf, xf[nt], zero4());
          t[0] += bv4.x; t[1] += bv4.y; t[2] += bv4.z; t[3] += bv4.w;
          if (node < a.N) {
            const int64_t row = (int64_t)b * a.row_sb + (int64_t)node * a.row_sn;
            L::st4(gqkv + row * 3 * D + part * D + DH * h + 4 * g, t[0], t[1], t[2], t[3]);
          }
          if (part == 0) {
            qs[nt] = L::mk(t[0] * a.scale, t[1] * a.scale, t[2] * a.scale, t[3] * a.scale);
            if (a.tie_qk) kf[nt] = L::mk(t);   // K tied to Q: tiles beyond n_real are never used
          } else {
            kf[nt] = L::mk(t);
          }
        } else {
          // (node = 4g + r, c' = lq)
          f32x4 t = dot_row_ops<T, D>(xf[nt], wf, zero4());
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            t[r] += bv1;
            const int nd = 16 * nt + 4 * g + r;
            if (nd < a.N) {
              const int64_t row = (int64_t)b * a.row_sb + (int64_t)nd * a.row_sn;
              L::st1(gqkv + row * 3 * D + 2 * D + DH * h + lq, t[r]);
            }
            if (nd >= n) t[r] = 0.0f;  // padded keys carry no value
          }
          vbo[nt] = L::mk(t);
        }
      }
    }
  }

  FETA_STAMP(2);
  // ---- attention core per 16-query tile (same arithmetic as attn_fwd_dense_kernel) ---------------
  const int bh = b * kBlkH + h;
  float* stg = scr + h * 16 * KP;
  // All query tiles advance together through each phase (scores, row max, exp / row sum, P.V): the
  // tiles are independent, so their MFMA chains and shuffle reductions interleave instead of running
  // one after the other.  Key tiles beyond n_real are skipped by wave-uniform branches.
  f32x4 acc[NT][NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int qb = 0; qb < NT; ++qb) acc[qb][kt] = zero4();
    if (16 * kt < n) {
#pragma unroll
      for (int qb = 0; qb < NT; ++qb) acc[qb][kt] = L::mma(kf[kt], qs[qb], zero4());  // (key 4g+r, query lq)
    }
  }
  float mx[NT], zs[NT], rinv[NT];
#pragma unroll
  for (int qb = 0; qb < NT; ++qb) {
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * kt + 4 * g + r < n) m = fmaxf(m, acc[qb][kt][r]);
    mx[qb] = m;
  }
#pragma unroll
  for (int qb = 0; qb < NT; ++qb) mx[qb] = fmaxf(mx[qb], shfl_xor(mx[qb], 16));
#pragma unroll
  for (int qb = 0; qb < NT; ++qb) mx[qb] = fmaxf(mx[qb], shfl_xor(mx[qb], 32));
#pragma unroll
  for (int qb = 0; qb < NT; ++qb) zs[qb] = 0.0f;
  // pe of this lane's (query lq, keys 4g .. 4g+3) pairs: one 16-byte LDS read per tile pair (columns >= N of the tile
  // hold whatever was there: those keys are masked by selects)
  float pv[NT][NT][4];
#pragma unroll
  for (int qb = 0; qb < NT; ++qb) {
    const int qc = min(16 * qb + lq, a.N - 1);
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const float4 t = *reinterpret_cast<const float4*>(Pe + qc * PEP + 16 * kt + 4 * g);
      pv[qb][kt][0] = t.x; pv[qb][kt][1] = t.y; pv[qb][kt][2] = t.z; pv[qb][kt][3] = t.w;
    }
  }
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    if (16 * kt >= n) continue;
#pragma unroll
    for (int qb = 0; qb < NT; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool kok = 16 * kt + 4 * g + r < n;
        const float e = kok ? fast_exp(acc[qb][kt][r] - mx[qb]) * pv[qb][kt][r] : 0.0f;   // (pv: see below)
        acc[qb][kt][r] = e;
        zs[qb] += e;
      }
  }
#pragma unroll
  for (int qb = 0; qb < NT; ++qb) zs[qb] += shfl_xor(zs[qb], 16);
#pragma unroll
  for (int qb = 0; qb < NT; ++qb) zs[qb] += shfl_xor(zs[qb], 32);
  f32x4 o[NT];
#pragma unroll
  for (int qb = 0; qb < NT; ++qb) {
    rinv[qb] = 1.0f / fmaxf(zs[qb], 1e-6f);
    o[qb] = zero4();
    const int q = 16 * qb + lq;
    if (g == 0 && q < a.N) {
      float* st = a.attn_stats + ((int64_t)bh * a.N + q) * 2;
      st[0] = mx[qb];
      st[1] = zs[qb];
    }
  }
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    if (16 * kt >= n) continue;
#pragma unroll
    for (int qb = 0; qb < NT; ++qb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[qb][kt][r] *= rinv[qb];
      o[qb] = L::mma(L::mk(acc[qb][kt]), vbo[kt], o[qb]);  // (query 4g+r, c' lq)
    }
  }
#pragma unroll
  for (int qb = 0; qb < NT; ++qb)
#pragma unroll
    for (int r = 0; r < 4; ++r) L::st1(Os + (16 * qb + 4 * g + r) * P + DH * h + lq, o[qb][r]);
  if (a.attn != nullptr) {
#pragma unroll
    for (int qb = 0; qb < NT; ++qb) {
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[lq * KP + 16 * kt + 4 * g + r] = acc[qb][kt][r];
      wave_lds_sync();
      const int rows = min(16, a.N - 16 * qb);
      float* dst = a.attn + ((int64_t)bh * a.N + 16 * qb) * a.N;
      for (int i = lane; i < rows * a.N; i += 64) {
        const int qq = i / a.N, kk = i - qq * a.N;
        dst[i] = stg[qq * KP + kk];
      }
      wave_lds_sync();
    }
  }
  FETA_STAMP(3);
  __syncthreads();
  FETA_STAMP(4);

  // ---- concat to HBM (whole rows) and out_proj: wave w owns output columns 16w .. 16w+15 ----------
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = tid + kRowThreads * i, node = idx / RV, q = idx % RV;
    if (node < a.N) {
      const int64_t row = (int64_t)b * a.row_sb + (int64_t)node * a.row_sn;
      const Vec ov = L::ldv(Os + node * P + L::VEC * q);
      L::stv(gout + row * D + L::VEC * q, ov);
      if (a.out_f32 != nullptr) {   // the same rows as fp32: operand of the fp32 filter stage behind a bf16 stack
        float f[L::VEC];
        L::unpack(ov, f);
#pragma unroll
        for (int e = 0; e < L::VEC; e += 4)
          *reinterpret_cast<float4*>(a.out_f32 + row * D + L::VEC * q + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
      }
    }
  }
  {
    RowOp<T, D> wf;
    load_row_op<T, D>(wf, Wo + (DH * h + lq) * P, g);
    const int o0 = DH * h + 4 * g;
    float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // shift of the statistics (feta_rowops.h): the running mean of the BatchNorm that will normalise y, as it is now
    float4 ks = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a.y_shift != nullptr) ks = *reinterpret_cast<const float4*>(a.y_shift + o0);
    const float kv[4] = {ks.x, ks.y, ks.z, ks.w};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (16 * nt >= a.N) break;
      const int node = 16 * nt + lq;
      const bool rok = node < a.N;
      const int64_t row = (int64_t)b * a.row_sb + (int64_t)min(node, a.N - 1) * a.row_sn;
      const float rs = rsv[nt];
      RowOp<T, D> of;
      load_row_op<T, D>(of, Os + node * P, g);
      const f32x4 t = dot_row_ops<T, D>(wf, of, zero4());  // (o = 16h + 4g + r, node lq)
      float res[4];
      L::ld4(Xs + node * P + o0, res);
      float v[4] = {(t[0] + bo.x) * rs + res[0], (t[1] + bo.y) * rs + res[1], (t[2] + bo.z) * rs + res[2],
                    (t[3] + bo.w) * rs + res[3]};
      if (rok) L::st4(gy + row * D + o0, v[0], v[1], v[2], v[3]);
      if (a.y_stats != nullptr) {   // (NULL: LayerNorm stack - nobody needs column statistics)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x1 = rok ? v[r] - kv[r] : 0.0f;
          s1[r] += row16_sum(x1);
          s2[r] += row16_sum(x1 * x1);
        }
      }
    }
    if (lq == 0 && a.y_stats != nullptr) {
      float* st = a.y_stats + (int64_t)b * 2 * D;
      *reinterpret_cast<float4*>(st + o0) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(st + D + o0) = make_float4(s2[0], s2[1], s2[2], s2[3]);
      if (b == 0) *reinterpret_cast<float4*>(a.y_stats + (int64_t)a.B * 2 * D + o0) = ks;   // row B: the shift
    }
  }
  }  // graphs of this workgroup
  FETA_STAMP(5);
  FETA_RT_LAUNCH_DONE(feta_block_launch);
}

// ---- round 3: eight waves per workgroup, optionally two workgroups per graph ------------------------------------
// The four-wave kernel above puts ONE wave on each SIMD of ONE CU per graph: nothing issues under that wave's 32-cycle
// fp32 MFMAs, the VALU work of a phase waits for them, and at the BASELINE batch (128 graphs) half of the 256 CUs have no
// graph at all.  Here a graph has S = 2 * WGS "query slots" - (workgroup w of the graph, wave
// parity p), slot = 2 w + p - and wave (head h, parity p) of workgroup w
//   * projects K (p = 0) or V (p = 1) of its head for ALL key tiles and hands it to its sibling through LDS (the tiles
//     leave the MFMA in the operand layout of the score / P.V products, so a lane writes and reads back 16 bytes per
//     tile: no conflicts, no shuffles); with two workgroups per graph both project every K and V tile (a 16-row tile
//     costs 2 x 16 MFMAs) and each writes its share of them to HBM;
//   * projects Q, runs the attention core and out_proj for the query tiles qb = slot (mod S) only;
//   * y's BatchNorm partial sums: the two parities of a head own the same 16 output columns over different rows and meet in
//     LDS; a workgroup emits ONE partial row (row b * WGS + w of y_stats: rows are complete per workgroup, nothing is
//     exchanged between workgroups).
// Every loop over query tiles stays a compile-time loop over all NT tiles with a wave-uniform ownership branch, so that
// register arrays are indexed statically (a runtime index would put them in scratch).
constexpr int kBlk8Threads = 512;

template <class T>
__host__ __device__ inline int block8_region_floats(int nt) {   // exchange region of one (head, tensor) = one wave's staging
  const int stg = 16 * (16 * nt + 1), xch = nt * 64 * (int)sizeof(typename Lp<T>::Op) / 4;
  return stg > xch ? stg : xch;
}

template <class T>
__host__ __device__ inline int block8_lds_bytes(int nt) {
  const int nr = 16 * nt, P = kBlkD + Lp<T>::PAD;
  int b = (int)sizeof(T) * (4 * kBlkD * P + 2 * nr * P);   // W_in, W_out, X tile, OUT tile
  int f = 2 * kBlkD + 128;                                 // scale / shift of the input BatchNorm; statistics hand-over
  const int fin = reduce_scratch_floats(kBlkD, kBlk8Threads), reg = 8 * block8_region_floats<T>(nt);
  f += fin > reg ? fin : reg;
  f += nr * (nr + 4);                                      // pe tile
  const int role = 4 * colsum_role_lds_floats(kBlk8Threads);
  const int tot = b + 4 * f;
  return tot > role ? tot : role;
}

template <class T, int NT, int WGS>
__global__ __launch_bounds__(kBlk8Threads) void attn_block_fwd8_kernel(BlockArgs a, ColsumPlan sums, int main_grid) {
  typedef Lp<T> L;
  typedef typename L::Op Op;
  typedef typename L::Vec Vec;
  constexpr int TH = kBlk8Threads;
  constexpr int D = kBlkD, DH = kBlkDH, P = kBlkD + L::PAD, NR = 16 * NT, KP = NR + 1;
  constexpr int S = 2 * WGS;                 // query slots of a graph
  constexpr int NQ = (NT + S - 1) / S;       // query tiles of a wave
  constexpr int KVU = NT >= 4 ? 2 : NT;      // K / V tiles in flight (their results go to LDS: the loop need not be unrolled, and at
                                             // four tiles the fully unrolled form spills)
  constexpr int RV = D / L::VEC;
  constexpr int XI = (NR * RV + TH - 1) / TH;
  constexpr int QR = WGS == 1 ? NR : (NR < 32 ? NR : 32);   // query rows of a workgroup (their pe rows are staged)
  constexpr int PEI = (QR * NR + TH - 1) / TH;
  if ((int)blockIdx.x >= main_grid) {
    colsum_role<TH>(sums, (int)blockIdx.x - main_grid);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = wv & 3, p = wv >> 2, lq = lane & 15, g = lane >> 4;
  const int gp = main_grid / WGS;            // graphs in flight; the workgroups of graph b are b, b + gp (one XCD when gp % 8 == 0)
  const int w = (int)blockIdx.x / gp, b0 = (int)blockIdx.x % gp;
  const int slot = 2 * w + p;
  T* Wi = reinterpret_cast<T*>(lds_bytes());   // [192][P]
  T* Wo = Wi + 3 * D * P;                      // [64][P]
  T* Xs = Wo + D * P;                          // [NR][P]
  T* Os = Xs + NR * P;                         // [NR][P]
  float* xss = reinterpret_cast<float*>(Os + NR * P);   // [2][64]
  float* sx = xss + 2 * D;                     // [4 heads][4 g][8]: statistics of the odd-parity waves
  float* scr = sx + 128;                       // finalize scratch, then the K / V hand-over, then the probability staging
  const int PER = block8_region_floats<T>(NT);
  Op* XK = reinterpret_cast<Op*>(scr + (2 * h) * PER);       // K^T tiles of head h (operand layout)
  Op* XV = reinterpret_cast<Op*>(scr + (2 * h + 1) * PER);   // V tiles of head h
  float* stg = scr + (2 * h + p) * PER;                      // this wave's probability staging (attn write)
  constexpr int PEP = NR + 4;
  float* Pe = reinterpret_cast<float*>(lds_bytes() + block8_lds_bytes<T>(NT)) - NR * PEP;
  const T* gx = reinterpret_cast<const T*>(a.x);
  const T* gpe = reinterpret_cast<const T*>(a.pe);
  T* gqkv = reinterpret_cast<T*>(a.qkv);
  T* gout = reinterpret_cast<T*>(a.out);
  T* gy = reinterpret_cast<T*>(a.y);
  const bool x_norm = a.x_stats != nullptr || a.x_bn != nullptr;
  FETA_STAMP(0);

  PartialBatchT<16> pb;   // (all 512 threads: 16 slices x 16 rows = every partial row of the BASELINE batch in one batch)
  partials_request_t<TH, 16>(a.x_stats != nullptr ? a.x_stats : a.w_in, a.x_stats != nullptr ? a.Gx : 0, D, pb);
  const bool has_pe = a.pe != nullptr;
  // operands of the graph at hand as they arrive from memory (storage type: nothing depends on them until they are staged),
  // requested while the graph before it is computed - a workgroup walks 4 graphs at config 5's batch, one workgroup per CU:
  // nobody else hides the latency
  Vec xv[XI];
  T pel[PEI];
  float rsn[NQ], rsv[NQ];
  int n = 0, n_req = 0;
  const int nn = a.N * a.N;
  // this workgroup's query rows [q0, q1) and their pe elements [q0 N, q1 N) - one contiguous stream
  const int q0 = WGS == 1 ? 0 : 32 * w, q1 = WGS == 1 ? a.N : min(a.N, 32 * w + 32);
  const int pe0 = q0 * a.N, pecnt = max(q1 - q0, 0) * a.N;
  auto request_graph = [&](int b) {
    n_req = a.n_real[b];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int idx = min(tid + TH * i, NR * RV - 1), node = idx / RV, q = idx % RV;
      const int64_t row = (int64_t)b * a.row_sb + (int64_t)min(node, a.N - 1) * a.row_sn;
      xv[i] = L::ldv(gx + row * D + L::VEC * q);
    }
#pragma unroll
    for (int i = 0; i < PEI; ++i)
      pel[i] = has_pe ? gpe[(int64_t)b * nn + pe0 + max(min(tid + TH * i, pecnt - 1), 0)] : L::one();
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int qc = min(16 * (slot + S * i) + lq, a.N - 1);
      rsn[i] = a.rowscale != nullptr ? a.rowscale[(int64_t)b * a.row_sb + (int64_t)qc * a.row_sn] : 1.0f;
    }
  };
  request_graph(b0);
  float4 binq = make_float4(0.0f, 0.0f, 0.0f, 0.0f), bink = binq, bo = binq;
  float bin1 = 0.0f;
  if (a.b_in != nullptr) {
    binq = *reinterpret_cast<const float4*>(a.b_in + DH * h + 4 * g);
    bink = *reinterpret_cast<const float4*>(a.b_in + (a.tie_qk ? 0 : D) + DH * h + 4 * g);
    bin1 = a.b_in[2 * D + DH * h + lq];
  }
  if (a.b_out != nullptr) bo = *reinterpret_cast<const float4*>(a.b_out + DH * h + 4 * g);
  float4 wvv[8];   // 256 rows of 16 float4 (W_in then W_out: fp32 masters), 8 per thread
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + TH * i, r = idx >> 4, q = idx & 15;
    const float* src = r < 3 * D ? a.w_in + (int64_t)r * D : a.w_out + (int64_t)(r - 3 * D) * D;
    wvv[i] = *reinterpret_cast<const float4*>(src + 4 * q);
  }
  float xg = 1.0f, xb = 0.0f, xk = 0.0f;
  if (a.x_stats != nullptr && tid < D) {
    xg = a.x_gamma[tid];
    xb = a.x_beta[tid];
    xk = partials_shift(a.x_stats, a.Gx, D, tid);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + TH * i;
    L::st4(Wi + (idx >> 4) * P + 4 * (idx & 15), wvv[i].x, wvv[i].y, wvv[i].z, wvv[i].w);
  }
  FETA_STAMP(6);
  if (a.x_stats != nullptr) {
    reduce_partials_finish_t<TH, 16>(a.x_stats, a.Gx, D, pb, scr + 2 * D, scr);
    for (int c = tid; c < D; c += TH) {
      float mean, var;
      bn_moments_k(xk, D, a.M, scr, c, mean, var);   // (c == tid: the loop runs once for the first D threads)
      const float rstd = rsqrtf(var + a.eps);
      const float scale = xg * rstd;
      const float shift = xb - mean * scale;
      xss[c] = scale;
      xss[D + c] = shift;
      if (blockIdx.x == 0) {
        a.x_bn_out[c] = scale;
        a.x_bn_out[D + c] = shift;
        a.x_bn_out[2 * D + c] = mean;
        a.x_bn_out[3 * D + c] = rstd;
        if (a.x_rmean != nullptr) {
          const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
          a.x_rmean[c] = (1.0f - a.momentum) * a.x_rmean[c] + a.momentum * mean;
          a.x_rvar[c] = (1.0f - a.momentum) * a.x_rvar[c] + a.momentum * unbiased;
        }
        if (c == 0 && a.x_nbt != nullptr) *a.x_nbt += 1;
      }
    }
  } else if (a.x_bn != nullptr) {
    for (int c = tid; c < 2 * D; c += TH) xss[c] = a.x_bn[c];
  } else if (a.x_ln_gamma != nullptr) {   // LayerNorm on load (feta_ln.h): xss = gamma | beta
    for (int c = tid; c < 2 * D; c += TH) xss[c] = c < D ? a.x_ln_gamma[c] : a.x_ln_beta[c - D];
  }
  const bool x_ln = a.x_ln_gamma != nullptr;
  FETA_STAMP(7);
  const int lane0 = lane;
  // column statistics of y: ONE partial row per workgroup - a workgroup that walks several graphs (B beyond the grid cap)
  // keeps adding to these registers (the sums are relative to the same shift) and stores once behind the loop; a row per
  // graph was 1024 rows at config 5, which the host then reduced with an extra launch per layer (fused_stack._cap_partials)
  float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int b = b0; b < a.B; b += gp) {
  // (LDS-only barriers from here on: the tiles of the previous graph have been consumed / xss is LDS data - the requests
  // of the next graph stay in flight across them)
  n = n_req;
#pragma unroll
  for (int i = 0; i < NQ; ++i) rsv[i] = rsn[i];
  // (everything a lane derives from its id is invariant in the graph loop and would be hoisted and held across the whole
  // body - csrc/block_bwd.hip: the lane id is laundered once per graph)
  int lane_l = lane0;
  FETA_OPAQUE_LANE(lane_l);
  const int lane = lane_l, tid = (wv << 6) | lane, lq = lane & 15, g = lane >> 4;
  lds_barrier();
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = tid + TH * i, node = idx / RV, q = idx % RV;
    if (XI * TH != NR * RV && idx >= NR * RV) continue;
    Vec v = xv[i];
    if (x_norm) {
      float f[L::VEC];
      L::unpack(v, f);
#pragma unroll
      for (int e = 0; e < L::VEC; ++e) f[e] = f[e] * xss[L::VEC * q + e] + xss[D + L::VEC * q + e];
      v = L::pack(f);
    } else if (x_ln) {
      // the row is held by RV consecutive lanes: mean / rstd by DPP sums over them, then gamma / beta (a wave-uniform
      // branch: a wave stages whole rows)
      float f[L::VEC];
      L::unpack(v, f);
      ln_apply<L::VEC>(f, xss + L::VEC * q, xss + D + L::VEC * q, a.eps);
      v = L::pack(f);
    }
    L::stv(Xs + node * P + L::VEC * q, v);  // rows >= N: a copy of row N-1, never stored
  }
  {
    const float rn = 1.0f / (float)a.N;
#pragma unroll
    for (int i = 0; i < PEI; ++i) {
      const int idx = tid + TH * i, e = pe0 + idx;
      const int qq = (int)(((float)e + 0.5f) * rn), kk = e - qq * a.N;
      if (idx < pecnt) Pe[qq * PEP + kk] = L::to_f(pel[i]);
    }
  }
  if (b + gp < a.B) request_graph(b + gp);   // (wave-uniform; xv / pel have been consumed)
  lds_barrier();
  FETA_STAMP(1);

  // ---- in_proj: Q^T (scaled) of this wave's query tiles; K^T (p = 0) or V (p = 1) of every key tile -> LDS ----------
  Op qs[NQ], kf[NT], vbo[NT];
  {
    // (the row operand of a tile is read from LDS where it is used: holding all NT of them costs 16 NT registers, and two
    // waves share a SIMD's register file here)
    {
      RowOp<T, D> wf;
      load_row_op<T, D>(wf, Wi + (DH * h + lq) * P, g);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if ((nt & (S - 1)) != slot) continue;
        const int node = 16 * nt + lq;
        RowOp<T, D> xf;
        load_row_op<T, D>(xf, Xs + node * P, g);
        f32x4 t = dot_row_ops<T, D>(wf, xf, zero4());   // (c = 4g + r, node = lq)
        t[0] += binq.x; t[1] += binq.y; t[2] += binq.z; t[3] += binq.w;
        if (node < a.N) {
          const int64_t row = (int64_t)b * a.row_sb + (int64_t)node * a.row_sn;
          L::st4(gqkv + row * 3 * D + DH * h + 4 * g, t[0], t[1], t[2], t[3]);
        }
        qs[nt / S] = L::mk(t[0] * a.scale, t[1] * a.scale, t[2] * a.scale, t[3] * a.scale);
      }
    }
    if (p == 0) {
      RowOp<T, D> wf;
      load_row_op<T, D>(wf, Wi + ((a.tie_qk ? 0 : D) + DH * h + lq) * P, g);
#pragma unroll KVU
      for (int nt = 0; nt < NT; ++nt) {
        Op kk = L::zero();
        if (16 * nt < n) {   // a key tile without a real node: no K (wave-uniform)
          const int node = 16 * nt + lq;
          RowOp<T, D> xf;
          load_row_op<T, D>(xf, Xs + node * P, g);
          f32x4 t = dot_row_ops<T, D>(wf, xf, zero4());
          t[0] += bink.x; t[1] += bink.y; t[2] += bink.z; t[3] += bink.w;
          if (!a.tie_qk && node < a.N && (nt % WGS) == w) {
            const int64_t row = (int64_t)b * a.row_sb + (int64_t)node * a.row_sn;
            L::st4(gqkv + row * 3 * D + D + DH * h + 4 * g, t[0], t[1], t[2], t[3]);
          }
          kk = L::mk(t);
        }
        L::sto(XK + nt * 64 + lane, kk);
      }
    } else {
      RowOp<T, D> wf;
      load_row_op<T, D>(wf, Wi + (2 * D + DH * h + lq) * P, g);
#pragma unroll KVU
      for (int nt = 0; nt < NT; ++nt) {
        Op vv = L::zero();
        if (16 * nt < n) {
          RowOp<T, D> xf;
          load_row_op<T, D>(xf, Xs + (16 * nt + lq) * P, g);
          f32x4 t = dot_row_ops<T, D>(xf, wf, zero4());   // (node = 4g + r, c' = lq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            t[r] += bin1;
            const int nd = 16 * nt + 4 * g + r;
            if (nd < a.N && (nt % WGS) == w) {
              const int64_t row = (int64_t)b * a.row_sb + (int64_t)nd * a.row_sn;
              L::st1(gqkv + row * 3 * D + 2 * D + DH * h + lq, t[r]);
            }
            if (nd >= n) t[r] = 0.0f;  // padded keys carry no value
          }
          vv = L::mk(t);
        }
        L::sto(XV + nt * 64 + lane, vv);
      }
    }
  }
  lds_barrier();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    kf[nt] = L::ldo(XK + nt * 64 + lane);
    vbo[nt] = L::ldo(XV + nt * 64 + lane);
  }
  FETA_STAMP(2);

  // ---- attention core of this wave's query tiles (the arithmetic of the four-wave kernel) -----------------------
  const int bh = b * kBlkH + h;
  // (scalars, not f32x4 acc[NQ][NT]: at 32 floats the optimizer promotes such an array to ONE vector value and every
  // conditional tile update copies the whole tuple - see csrc/attnout.hip)
  float acc[NQ][NT][4];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      f32x4 t = zero4();
      if (16 * kt < n && slot + S * i < NT) t = L::mma(kf[kt], qs[i], zero4());  // (key 4g+r, query lq)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][kt][r] = t[r];
    }
  }
  float mx[NQ], zs[NQ], rinv[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * kt + 4 * g + r < n) m = fmaxf(m, acc[i][kt][r]);
    mx[i] = m;
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) mx[i] = fmaxf(mx[i], shfl_xor(mx[i], 16));
#pragma unroll
  for (int i = 0; i < NQ; ++i) mx[i] = fmaxf(mx[i], shfl_xor(mx[i], 32));
#pragma unroll
  for (int i = 0; i < NQ; ++i) zs[i] = 0.0f;
  // pe of this lane's (query lq, keys 4g .. 4g+3) pairs: one 16-byte LDS read per tile pair, where it is used
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    if (16 * kt >= n) continue;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if (slot + S * i >= NT) continue;
      const int qc = min(16 * (slot + S * i) + lq, a.N - 1);
      const float4 t = *reinterpret_cast<const float4*>(Pe + qc * PEP + 16 * kt + 4 * g);
      const float pv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool kok = 16 * kt + 4 * g + r < n;
        const float e = kok ? fast_exp(acc[i][kt][r] - mx[i]) * pv[r] : 0.0f;
        acc[i][kt][r] = e;
        zs[i] += e;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) zs[i] += shfl_xor(zs[i], 16);
#pragma unroll
  for (int i = 0; i < NQ; ++i) zs[i] += shfl_xor(zs[i], 32);
  f32x4 o[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    rinv[i] = 1.0f / fmaxf(zs[i], 1e-6f);
    o[i] = zero4();
    const int qb = slot + S * i, q = 16 * qb + lq;
    if (g == 0 && qb < NT && q < a.N) {
      float* st = a.attn_stats + ((int64_t)bh * a.N + q) * 2;
      st[0] = mx[i];
      st[1] = zs[i];
    }
  }
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    if (16 * kt >= n) continue;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if (slot + S * i >= NT) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][kt][r] *= rinv[i];
      o[i] = L::mma(L::mk(acc[i][kt][0], acc[i][kt][1], acc[i][kt][2], acc[i][kt][3]), vbo[kt], o[i]);  // (query 4g+r, c' lq)
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int qb = slot + S * i;
    if (qb < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) L::st1(Os + (16 * qb + 4 * g + r) * P + DH * h + lq, o[i][r]);
    }
  }
  if (a.attn != nullptr) {
    lds_barrier();   // the staging area of this wave is the K / V hand-over its sibling may still be reading
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int qb = slot + S * i;
      if (qb >= NT) continue;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[lq * KP + 16 * kt + 4 * g + r] = acc[i][kt][r];
      wave_lds_sync();
      const int rows = min(16, a.N - 16 * qb);
      float* dst = a.attn + ((int64_t)bh * a.N + 16 * qb) * a.N;
      for (int j = lane; j < rows * a.N; j += 64) {
        const int qq = j / a.N, kk = j - qq * a.N;
        dst[j] = stg[qq * KP + kk];
      }
      wave_lds_sync();
    }
  }
  FETA_STAMP(3);
  lds_barrier();
  FETA_STAMP(4);

  // ---- concat rows of this workgroup's tiles to HBM; out_proj: wave (h, p) owns output columns 16h .. 16h+15 of its
  // own query tiles
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = tid + TH * i, node = idx / RV, q = idx % RV;
    if (node < a.N && (((node >> 4) & (S - 1)) >> 1) == w) {
      const int64_t row = (int64_t)b * a.row_sb + (int64_t)node * a.row_sn;
      const Vec ov = L::ldv(Os + node * P + L::VEC * q);
      L::stv(gout + row * D + L::VEC * q, ov);
      if (a.out_f32 != nullptr) {
        float f[L::VEC];
        L::unpack(ov, f);
#pragma unroll
        for (int e = 0; e < L::VEC; e += 4)
          *reinterpret_cast<float4*>(a.out_f32 + row * D + L::VEC * q + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
      }
    }
  }
  {
    RowOp<T, D> wf;
    load_row_op<T, D>(wf, Wo + (DH * h + lq) * P, g);
    const int o0 = DH * h + 4 * g;
    float4 ks = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a.y_shift != nullptr) ks = *reinterpret_cast<const float4*>(a.y_shift + o0);
    const float kv[4] = {ks.x, ks.y, ks.z, ks.w};
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int qb = slot + S * i;
      if (qb >= NT) continue;
      const int node = 16 * qb + lq;
      const bool rok = node < a.N;
      const int64_t row = (int64_t)b * a.row_sb + (int64_t)min(node, a.N - 1) * a.row_sn;
      const float rs = rsv[i];
      RowOp<T, D> of;
      load_row_op<T, D>(of, Os + node * P, g);
      const f32x4 t = dot_row_ops<T, D>(wf, of, zero4());  // (o = 16h + 4g + r, node lq)
      float res[4];
      L::ld4(Xs + node * P + o0, res);
      float v[4] = {(t[0] + bo.x) * rs + res[0], (t[1] + bo.y) * rs + res[1], (t[2] + bo.z) * rs + res[2],
                    (t[3] + bo.w) * rs + res[3]};
      if (rok) L::st4(gy + row * D + o0, v[0], v[1], v[2], v[3]);
      if (a.y_stats != nullptr) {   // (NULL: LayerNorm stack - nobody needs column statistics)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x1 = rok ? v[r] - kv[r] : 0.0f;
          s1[r] += row16_sum(x1);
          s2[r] += row16_sum(x1 * x1);
        }
      }
    }
  }
  }  // graphs of this workgroup
  if (a.y_stats != nullptr) {
    // the two parities of a head hold sums over different rows of the same columns: odd hands over, even adds and stores
    const int o0 = DH * h + 4 * g;
    if (p == 1 && lq == 0) {
      float* e = sx + (4 * h + g) * 8;
      *reinterpret_cast<float4*>(e) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(e + 4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    }
    lds_barrier();
    if (p == 0 && lq == 0) {
      const float* e = sx + (4 * h + g) * 8;
      const float4 t1 = *reinterpret_cast<const float4*>(e), t2 = *reinterpret_cast<const float4*>(e + 4);
      float* st = a.y_stats + ((int64_t)b0 * WGS + w) * 2 * D;
      *reinterpret_cast<float4*>(st + o0) = make_float4(s1[0] + t1.x, s1[1] + t1.y, s1[2] + t1.z, s1[3] + t1.w);
      *reinterpret_cast<float4*>(st + D + o0) = make_float4(s2[0] + t2.x, s2[1] + t2.y, s2[2] + t2.z, s2[3] + t2.w);
      if (b0 == 0 && w == 0) {   // the shift row, behind the gp * WGS partial rows
        float4 ks = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (a.y_shift != nullptr) ks = *reinterpret_cast<const float4*>(a.y_shift + o0);
        *reinterpret_cast<float4*>(a.y_stats + (int64_t)gp * WGS * 2 * D + o0) = ks;
      }
    }
  }
  FETA_STAMP(5);
  FETA_RT_LAUNCH_DONE(feta_block_launch);
}

// Form of a forward launch: waves per workgroup and workgroups per graph.  Two workgroups per graph where a graph has
// three or four query tiles and the launch still fits the chip (one resident workgroup per CU); FETA_BLOCK_FWD_WAVES=4
// selects the four-wave kernel of rounds 1-2, FETA_BLOCK_FWD_WGS=1|2 overrides the split (A/B timing, tests).
struct BlockFwdForm {
  int waves, wgs, cap;
};
static BlockFwdForm block_fwd_form(int B, int N) {
  BlockFwdForm f{8, 1, kBlkMaxGrid};
  if (const char* e = getenv("FETA_BLOCK_MAX_GRID")) f.cap = atoi(e) > 0 ? atoi(e) : f.cap;
  if (const char* e = getenv("FETA_BLOCK_FWD_WAVES")) f.waves = atoi(e) == 4 ? 4 : 8;
  if (f.waves == 4) return f;
  const int nt = (N + 15) / 16;
  f.wgs = (nt >= 3 && 2 * B <= f.cap) ? 2 : 1;
  if (const char* e = getenv("FETA_BLOCK_FWD_WGS")) {
    const int v = atoi(e);
    if (v == 1 || (v == 2 && nt >= 3)) f.wgs = v;   // (two workgroups need query tiles for both)
  }
  return f;
}

template <class T, int NT, int WGS>
int launch_block_fwd8(const BlockArgs& a, const feta_colsum_seg* segs, int nseg, int cap, hipStream_t stream) {
  const size_t lds = block8_lds_bytes<T>(NT);
  auto kern = attn_block_fwd8_kernel<T, NT, WGS>;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  int gp = cap / WGS;   // graphs in flight
  if (gp < 1) gp = 1;
  if (a.B < gp) gp = a.B;
  const int grid = gp * WGS;
  ColsumPlan plan{};
  const int tiles = plan_colsum(segs, nseg, plan);
  hipLaunchKernelGGL(kern, dim3(grid + tiles), dim3(kBlk8Threads), lds, stream, a, plan, grid);
  return check_launch("feta_attn_block_fwd");
}

template <class T, int NT>
int launch_block_fwd(const BlockArgs& a, const feta_colsum_seg* segs, int nseg, hipStream_t stream) {
  const size_t lds = block_lds_bytes<T>(NT, a.attn != nullptr);
  auto kern = attn_block_fwd_kernel<T, NT>;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  int cap = kBlkMaxGrid;   // one resident workgroup per CU (LDS); FETA_BLOCK_MAX_GRID: tests force the loop
  if (const char* e = getenv("FETA_BLOCK_MAX_GRID")) cap = atoi(e) > 0 ? atoi(e) : cap;
  const int grid = a.B < cap ? a.B : cap;
  ColsumPlan plan{};
  const int tiles = plan_colsum(segs, nseg, plan);
  // Measured (round 3, B = 128): 0.2812 vs 0.2803 ms/step fp32, 0.2385 vs 0.2399 bf16, the phase stamps sum to the same
  // 4.7 us either way - the prologue is a chain of LDS work and barriers, not a wait for the weight stream.  Off by
  // default (and the fp32 four-tile instantiation would spill 92 B per lane holding the weights that long).
  int weights_last = 0;
  if (const char* e = getenv("FETA_BLOCK_WEIGHTS_LAST")) weights_last = atoi(e) != 0 && !(NT == 4 && sizeof(T) == sizeof(float));
  hipLaunchKernelGGL(kern, dim3(grid + tiles), dim3(kRowThreads), lds, stream, a, plan, grid, weights_last);
  return check_launch("feta_attn_block_fwd");
}

template <class T>
int dispatch_block_fwd(const BlockArgs& a, const feta_colsum_seg* segs, int nseg, hipStream_t stream) {
  const BlockFwdForm f = block_fwd_form(a.B, a.N);
  if (f.waves == 8) {
    if (f.wgs == 2) {
      if ((a.N + 15) / 16 == 3) return launch_block_fwd8<T, 3, 2>(a, segs, nseg, f.cap, stream);
      return launch_block_fwd8<T, 4, 2>(a, segs, nseg, f.cap, stream);
    }
    switch ((a.N + 15) / 16) {
      case 1: return launch_block_fwd8<T, 1, 1>(a, segs, nseg, f.cap, stream);
      case 2: return launch_block_fwd8<T, 2, 1>(a, segs, nseg, f.cap, stream);
      case 3: return launch_block_fwd8<T, 3, 1>(a, segs, nseg, f.cap, stream);
      default: return launch_block_fwd8<T, 4, 1>(a, segs, nseg, f.cap, stream);
    }
  }
  switch ((a.N + 15) / 16) {
    case 1: return launch_block_fwd<T, 1>(a, segs, nseg, stream);
    case 2: return launch_block_fwd<T, 2>(a, segs, nseg, stream);
    case 3: return launch_block_fwd<T, 3>(a, segs, nseg, stream);
    default: return launch_block_fwd<T, 4>(a, segs, nseg, stream);
  }
}

}  // namespace feta

using namespace feta;

#ifdef FETA_TIMING
extern "C" int feta_debug_block_stamps(unsigned long long* out256) {
  return (int)hipMemcpyFromSymbol(out256, HIP_SYMBOL(feta_block_stamps), sizeof(unsigned long long) * 256);
}
#endif

extern "C" int feta_attn_block_supported(int N, int d_model, int heads) {
  return (d_model == kBlkD && heads == kBlkH && N >= 1 && N <= 64) ? 1 : 0;
}

extern "C" int feta_attn_block_stat_rows(int B, int N) {
  if (B < 1 || N < 1 || N > 64) return 0;
  const BlockFwdForm f = block_fwd_form(B, N);
  if (f.waves == 4) return B;                    // the four-wave kernel: a row per graph
  int gp = f.cap / f.wgs;                        // (launch_block_fwd8: graphs in flight)
  if (gp < 1) gp = 1;
  return (B < gp ? B : gp) * f.wgs;              // one row per workgroup
}

extern "C" int feta_attn_block_fwd(const feta_attn_block* d, feta_stream_t stream) {
  return feta_attn_block_fwd_sums(d, nullptr, 0, stream);
}

extern "C" int feta_attn_block_fwd_sums(const feta_attn_block* d, const feta_colsum_seg* segs, int nseg,
                                        feta_stream_t stream) {
  FETA_REQUIRE(d != nullptr, "attn_block_fwd: null descriptor");
  FETA_REQUIRE(nseg >= 0 && nseg <= FETA_COLSUM_MAX_SEGS && (nseg == 0 || segs != nullptr),
               "attn_block_fwd: 0..%d column-sum segments", FETA_COLSUM_MAX_SEGS);
  for (int i = 0; i < nseg; ++i) FETA_REQUIRE(colsum_seg_ok(segs[i]), "attn_block_fwd: bad segment %d", i);
  const BlockArgs& a = *d;
  FETA_REQUIRE(a.x && a.w_in && a.w_out && a.n_real && a.qkv && a.out && a.attn_stats && a.y,
               "attn_block_fwd: null pointer");
  FETA_REQUIRE(a.B > 0 && a.N >= 1 && a.N <= 64, "attn_block_fwd: N=%d outside [1,64]", a.N);
  FETA_REQUIRE(a.M == a.B * a.N, "attn_block_fwd: M=%d is not B*N", a.M);
  FETA_REQUIRE(a.x_stats == nullptr || (a.x_gamma && a.x_beta && a.x_bn_out && a.Gx > 0),
               "attn_block_fwd: x_stats needs x_gamma, x_beta, x_bn_out, Gx");
  FETA_REQUIRE(aligned16(a.x) && aligned16(a.w_in) && aligned16(a.w_out) && aligned16(a.qkv) && aligned16(a.out) &&
               aligned16(a.y) && aligned16(a.y_stats) && aligned16(a.x_stats) && aligned16(a.b_in) && aligned16(a.b_out) &&
               aligned16(a.out_f32) && aligned16(a.y_shift),
               "attn_block_fwd: tensors must be 16-byte aligned");
  FETA_REQUIRE(a.x_ln_gamma == nullptr || (a.x_ln_beta != nullptr && a.x_stats == nullptr && a.x_bn == nullptr),
               "attn_block_fwd: x_ln_gamma needs x_ln_beta and excludes x_bn / x_stats");
  FETA_REQUIRE(a.dtype == FETA_F32 || a.dtype == FETA_BF16, "attn_block_fwd: dtype %d", a.dtype);
  if (a.dtype == FETA_BF16) return dispatch_block_fwd<bf16_t>(a, segs, nseg, (hipStream_t)stream);
  return dispatch_block_fwd<float>(a, segs, nseg, (hipStream_t)stream);
}
