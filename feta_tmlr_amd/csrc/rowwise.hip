// Row-wise (per-node) part of the encoder layer (A1) and the combine (A4): the in/out
// projections, the FFN linears and linear_cat with their element-wise neighbours fused, and
// training-mode BatchNorm1d over all N*B rows.  Replaces, per DiffTransformerEncoderLayer
// call (contract transformer/models.py:166-167; body per upstream GraphiT, README.md:129):
//   [BatchNorm of the input] -> F.linear(+bias) -> [relu] -> [* degree] -> [+ residual]
//   in one launch, plus the BN statistics of the result as per-block partial sums;
//   backward: [BatchNorm backward of the incoming gradient] -> dX (+ residual gradient, +
//   partial sums for the next BatchNorm backward) and the split-K weight/bias gradient in ONE
//   launch (blocks take roles), partials reduced deterministically by colsum_kernel.
// BatchNorm never runs as its own pass on the fused path: its statistics are produced by the
// producer's epilogue, its apply happens in the consumers' operand loads ("a tensor seen
// through a BatchNorm" = pre-norm values + a [4][D] parameter block scale/shift/mean/rstd).
// Activations are [M, C] row-major with M = N*B rows (seq-first rows are contiguous).
//
// Decomposition: a workgroup = 64 rows (one 16-row block per wave) x a group of up to 4
// output tiles; the weight slice of the group is staged ONCE in LDS by all 256 threads
// (padded pitch, conflict-free operand reads) - the small problem is latency-bound, so the
// dependent global loads of the weights are what has to go.
// MFMA orientation: Y^T tile = W_tile . X_tile^T, so the accumulator holds 4 consecutive
// output features of one row per lane -> 16-byte epilogue loads/stores, and the row-wise
// epilogue operands (degree, residual) are lane-local.
#include <cstdlib>

#include "feta_abi_common.h"
#include "feta_tiles.h"
#include "feta_rowops.h"

namespace feta {

constexpr int kRowsPerBlock = 16 * kRowWaves;
constexpr int kMaxRowBlocks = 256;  // cap on per-block partial statistics
constexpr int kMaxChunks = 128;     // cap on split-K chunks of the weight gradient
constexpr int kChunkRowBlocks = 4;  // 16-row blocks per split-K chunk (one 64-row LDS tile)

typedef feta_rowlin_ex RowLinArgs;  // include/feta_hip.h

struct RowLinGeom {
  int G;          // row groups of the grid (row blocks are strided over it)
  int TG;         // output tiles per workgroup (forward) / k tiles per workgroup (dX role)
  int RC;         // weight-gradient row chunks
  int dx_blocks;  // backward: blocks [0, dx_blocks) compute dX, the rest dW/db partials
};

// x may be the virtual concatenation [x (first x_split columns) | x2 (the rest)] - linear_cat without
// materialising torch.cat (transformer/models.py:223).  x_split is a multiple of 16.
__device__ __forceinline__ const float* x_at(const RowLinArgs& a, int64_t row, int k) {
  if (a.x2 == nullptr) return a.x + row * a.KI + k;
  return k < a.x_split ? a.x + row * a.x_split + k : a.x2 + row * (a.KI - a.x_split) + (k - a.x_split);
}
__device__ __forceinline__ float* dx_at(const RowLinArgs& a, int64_t row, int k) {
  if (a.dx2 == nullptr) return a.dx + row * a.KI + k;
  return k < a.x_split ? a.dx + row * a.x_split + k : a.dx2 + row * (a.KI - a.x_split) + (k - a.x_split);
}

// ---- forward ------------------------------------------------------------------------------
template <int KI>
__global__ __launch_bounds__(kRowThreads) void rowlin_fwd_kernel(RowLinArgs a, RowLinGeom ge) {
  constexpr int LDW = KI + 4;  // LDS pitch of a weight row (16-byte aligned, bank-shifted)
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int rg = blockIdx.x % ge.G, og = blockIdx.x / ge.G;
  const int tgw = ge.TG * 16;
  const int o_base = og * tgw;
  const int ntile = min(ge.TG, a.NO / 16 - og * ge.TG);
  float* wt = feta_lds;                     // [TG*16][LDW]
  float* red = wt + tgw * LDW;              // [kRowWaves][2][TG*16]
  float* xss = red + kRowWaves * 2 * tgw;   // [2][KI] scale, shift of the input BatchNorm
  float* scr = xss + 2 * KI;                // finalize scratch
  const bool want_stats = a.stats != nullptr;
  const bool x_norm = a.x_stats != nullptr || a.x_bn != nullptr;

  // ---- load batch of a row block: everything it needs from memory is requested up front
  // (clamped row, selects afterwards), so the block pays ONE memory latency; the first batch is
  // issued before the weight staging / statistics finalize so that it overlaps with them
  struct Batch {
    Feat<KI> xf;
    float rs;
    float4 bv[4], rv[4], r1[4], r2[4];
  };
  auto load_batch = [&](int rb, Batch& B) {
    const int row = rb * kRowsPerBlock + wave_id() * 16 + lq;
    const int rowc = min(row, a.M - 1);
    if (a.x2 == nullptr) {
      load_row_sel<KI>(B.xf, a.x + (int64_t)rowc * KI, true, g);
    } else {
#pragma unroll
      for (int j = 0; j < Feat<KI>::NJ; ++j) {
        const int c = 16 * j + 4 * g;
        const float4 xv = *reinterpret_cast<const float4*>(x_at(a, rowc, c < KI ? c : 0));
        B.xf.f[j][0] = xv.x; B.xf.f[j][1] = xv.y; B.xf.f[j][2] = xv.z; B.xf.f[j][3] = xv.w;
      }
    }
    B.rs = a.rowscale != nullptr ? a.rowscale[rowc] : 1.0f;
    const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      B.bv[t] = z4; B.rv[t] = z4; B.r1[t] = make_float4(1.0f, 1.0f, 1.0f, 1.0f); B.r2[t] = z4;
      if (t < ntile) {
        const int o = o_base + 16 * t + 4 * g;
        if (a.bias != nullptr) B.bv[t] = *reinterpret_cast<const float4*>(a.bias + o);
        if (a.residual != nullptr)
          B.rv[t] = *reinterpret_cast<const float4*>(a.residual + (int64_t)rowc * a.NO + o);
        if (a.res_bn != nullptr) {
          B.r1[t] = *reinterpret_cast<const float4*>(a.res_bn + o);
          B.r2[t] = *reinterpret_cast<const float4*>(a.res_bn + a.NO + o);
        }
      }
    }
  };
  Batch cur;
  load_batch(rg, cur);

  const int nvec = ntile * 16 * (KI / 4);
  stage_float4(
      nvec,
      [&](int idx) {
        const int row = idx / (KI / 4), c4 = idx - row * (KI / 4);
        return reinterpret_cast<const float4*>(a.w + (int64_t)(o_base + row) * KI + 4 * c4);
      },
      [&](int idx) {
        const int row = idx / (KI / 4), c4 = idx - row * (KI / 4);
        return reinterpret_cast<float4*>(wt + row * LDW + 4 * c4);
      });
  if (want_stats)
    for (int i = threadIdx.x; i < kRowWaves * 2 * tgw; i += kRowThreads) red[i] = 0.0f;
  // a concatenated operand [x | x2] (linear_cat) is seen through its BatchNorm on the x part only: DS columns
  const int DS = a.x2 != nullptr ? a.x_split : KI;
  if (a.x2 != nullptr && x_norm)
    for (int c = DS + threadIdx.x; c < KI; c += kRowThreads) {
      xss[c] = 1.0f;
      xss[KI + c] = 0.0f;
    }
  if (a.x_stats != nullptr) {
    // first consumer of fresh statistics: finalize them (every block, redundantly and
    // deterministically); block 0 publishes the parameter block and the running statistics
    reduce_partials_t<kRowThreads, 32>(a.x_stats, a.Gx, DS, scr + 2 * DS, scr);   // (256 rows of 64 columns in ONE batch: feta_rowops.h)
    for (int c = threadIdx.x; c < DS; c += kRowThreads) {
      float mean, var;
      bn_moments(a.x_stats, a.Gx, DS, a.M, scr, c, mean, var);
      const float rstd = rsqrtf(var + a.eps);
      const float scale = a.x_gamma[c] * rstd;
      const float shift = a.x_beta[c] - mean * scale;
      xss[c] = scale;
      xss[KI + c] = shift;
      if (blockIdx.x == 0) {
        a.x_bn_out[c] = scale;
        a.x_bn_out[DS + c] = shift;
        a.x_bn_out[2 * DS + c] = mean;
        a.x_bn_out[3 * DS + c] = rstd;
        if (a.x_rmean != nullptr) {
          const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
          a.x_rmean[c] = (1.0f - a.momentum) * a.x_rmean[c] + a.momentum * mean;
          a.x_rvar[c] = (1.0f - a.momentum) * a.x_rvar[c] + a.momentum * unbiased;
        }
        if (c == 0 && a.x_nbt != nullptr) *a.x_nbt += 1;
      }
    }
  } else if (a.x_bn != nullptr) {
    for (int c = threadIdx.x; c < DS; c += kRowThreads) {
      xss[c] = a.x_bn[c];
      xss[KI + c] = a.x_bn[DS + c];
    }
  }
  __syncthreads();

  float* my = red + wave_id() * 2 * tgw;
  const int nrb = (a.M + kRowsPerBlock - 1) / kRowsPerBlock;
  for (int rb = rg; rb < nrb; rb += ge.G) {
    const int row = rb * kRowsPerBlock + wave_id() * 16 + lq;
    const bool rok = row < a.M;
    if (rb != rg) load_batch(rb, cur);
    Feat<KI>& xf = cur.xf;
    const float rs = cur.rs;
    float4 (&bv)[4] = cur.bv;
    float4 (&rv)[4] = cur.rv;
    float4 (&r1)[4] = cur.r1;
    float4 (&r2)[4] = cur.r2;
    if (x_norm) {
#pragma unroll
      for (int j = 0; j < Feat<KI>::NJ; ++j) {
        const int c = 16 * j + 4 * g;
        if (c < KI) {
#pragma unroll
          for (int s = 0; s < 4; ++s) xf.f[j][s] = xf.f[j][s] * xss[c + s] + xss[KI + c + s];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t >= ntile) break;
      Feat<KI> wf;
      load_row<KI>(wf, wt + (16 * t + lq) * LDW, g);
      f32x4 acc = dot_rows<KI>(wf, xf, zero4());  // (o = o_base + 16t + 4g + r, row)
      const int ol = 16 * t + 4 * g, o = o_base + ol;
      const float bb[4] = {bv[t].x, bv[t].y, bv[t].z, bv[t].w};
      const float rr[4] = {rv[t].x * r1[t].x + r2[t].x, rv[t].y * r1[t].y + r2[t].y,
                           rv[t].z * r1[t].z + r2[t].z, rv[t].w * r1[t].w + r2[t].w};
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[r] + bb[r];
        if (a.relu) v[r] = fmaxf(v[r], 0.0f);
        v[r] = v[r] * rs + (a.residual != nullptr ? rr[r] : 0.0f);
      }
      if (rok)
        *reinterpret_cast<float4*>(a.y + (int64_t)row * a.NO + o) = make_float4(v[0], v[1], v[2], v[3]);
      if (want_stats) {
        // (shifted sums, feta_rowops.h: relative to the consumer BatchNorm's running mean)
        float4 ks = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (a.stats_shift != nullptr) ks = *reinterpret_cast<const float4*>(a.stats_shift + o);
        const float kv[4] = {ks.x, ks.y, ks.z, ks.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s1 = rok ? v[r] - kv[r] : 0.0f, s2 = s1 * s1;
          s1 = row16_sum(s1);
          s2 = row16_sum(s2);
          if (lq == 0) {
            my[ol + r] += s1;
            my[tgw + ol + r] += s2;
          }
        }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * tgw; i += kRowThreads) {
      const int which = i / tgw, ol = i - which * tgw;
      if (ol < ntile * 16) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < kRowWaves; ++w) s += red[w * 2 * tgw + i];
        a.stats[((int64_t)rg * 2 + which) * a.NO + o_base + ol] = s;
        if (rg == 0 && which == 0)   // row G: the shift
          a.stats[(int64_t)ge.G * 2 * a.NO + o_base + ol] = a.stats_shift != nullptr ? a.stats_shift[o_base + ol] : 0.0f;
      }
    }
  }
}

#ifdef FETA_TIMING
__device__ unsigned long long feta_rowlin_stamps[32];
#endif
#define RL_STAMP_X(i) FETA_STAMP_TO(feta_rowlin_stamps, i, blockIdx.x == 0 && threadIdx.x == 0)
#define RL_STAMP_W(i) FETA_STAMP_TO(feta_rowlin_stamps, 16 + (i), (int)blockIdx.x == ge.dx_blocks && threadIdx.x == 0)

// ---- backward ---------------------------------------------------------------------------------
// gradient source g(row, o) = T(dy)(row, o) * rowscale[row] * [relu_y > 0], with T = identity or
// the BatchNorm backward  scale_o (dy - m1_o - xhat m2_o),  xhat = (g_y - mean_o) rstd_o,
// m1 = mean_rows(dy), m2 = mean_rows(dy xhat).  gv (LDS) = [5][NO]: scale, mean, rstd, m1, m2.

template <int NO>
__device__ void rowlin_dx_role(const RowLinArgs& a, const RowLinGeom& ge, const float* gv,
                               float* lds_free, int lq, int g) {
  // A workgroup = 64 rows x up to 4 k tiles.  The gradient tile g[64][NO] is staged ONCE in LDS with all
  // transforms applied (BatchNorm backward, relu mask, row scale); a wave owns ONE k tile: its weight
  // column slice W[:, k tile] sits in registers for the whole launch (NO/4 values per lane) and the
  // wave walks the row tiles with it, reading the gradient rows as 16-byte LDS operands.  Per MFMA that
  // is 1/4 of an LDS read instead of one, and a wave's columns are its own (no cross-wave sums unless
  // fewer than 4 k tiles leave waves free to split the rows).
  constexpr int GP = NO + 4, NJ = Feat<NO>::NJ;
  const int rg = blockIdx.x % ge.G, kg = blockIdx.x / ge.G;
  const int ks = ge.TG * 16;
  const int k_base = kg * ks;
  const int ntile = min(ge.TG, a.KI / 16 - kg * ge.TG);
  float* gt = lds_free;          // [64][GP]
  float* ev = gt + 64 * GP;      // [7][ks] epilogue vectors of this k slice:
  // add_bn scale, mean, rstd; add_fin m1, m2; sum_bn mean, rstd
  float* red = ev + 7 * ks;      // [kRowWaves][2][16] column sums of the waves
  const bool want_sums = a.sum_out != nullptr;
  const bool gbn = a.g_y != nullptr;
  // concatenated operand [x | x2]: sum_y / sum_bn / sum_out describe the BatchNorm of the x part (DS columns)
  const int DS = a.x2 != nullptr ? a.x_split : a.KI;
  for (int i = threadIdx.x; i < ks; i += kRowThreads) {
    const int k = k_base + i;
    const bool kok = k < a.KI;
    const bool ad = a.add_dout != nullptr && kok;
    ev[i] = ad ? a.add_bn[k] : 0.0f;
    ev[ks + i] = ad ? a.add_bn[2 * a.KI + k] : 0.0f;
    ev[2 * ks + i] = ad ? a.add_bn[3 * a.KI + k] : 0.0f;
    ev[3 * ks + i] = ad ? a.add_fin[k] : 0.0f;
    ev[4 * ks + i] = ad ? a.add_fin[a.KI + k] : 0.0f;
    ev[5 * ks + i] = (want_sums && k < DS) ? a.sum_bn[2 * DS + k] : 0.0f;
    ev[6 * ks + i] = (want_sums && k < DS) ? a.sum_bn[3 * DS + k] : 0.0f;
  }
  // wave -> (k tile, row tiles): 4 tiles: one tile, all 4 row tiles; 2 tiles: 2 row tiles; 1 tile: 1
  const int w = wave_id();
  const int groups = kRowWaves / ntile, per = 4 / (groups > 0 ? groups : 1);
  const bool active = w < groups * ntile;
  const int t = active ? w % ntile : 0, rt0 = active ? (w / ntile) * per : 0;
  const int kcol = k_base + 16 * t + lq;
  float wA[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int o = 16 * j + 4 * g + s;
      const float wv = a.w[(int64_t)(o < NO ? o : 0) * a.KI + (kcol < a.KI ? kcol : 0)];
      wA[j][s] = (o < NO && kcol < a.KI && active) ? wv : 0.0f;
    }
  RL_STAMP_X(1);
  float sum1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sum2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int nrb = (a.M + kRowsPerBlock - 1) / kRowsPerBlock;
  constexpr int gq = NO / 4;
  for (int rb = rg; rb < nrb; rb += ge.G) {
    const int r0 = rb * kRowsPerBlock;
    const int row_last = a.M - 1;
    if (rb != rg) __syncthreads();  // the previous tile has been consumed
    // stage the gradient tile: four items per thread per pass, every load of the pass issued first
    for (int base = threadIdx.x; base < 64 * gq; base += 4 * kRowThreads) {
      float4 dv[4], yv[4], rv[4];
      float rsv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = min(base + u * kRowThreads, 64 * gq - 1);
        const int rr = idx / gq, c4 = idx - rr * gq;
        const int64_t off = (int64_t)min(r0 + rr, row_last) * NO + 4 * c4;
        dv[u] = *reinterpret_cast<const float4*>(a.dy + off);
        if (gbn) yv[u] = *reinterpret_cast<const float4*>(a.g_y + off);
        if (a.relu_y != nullptr) rv[u] = *reinterpret_cast<const float4*>(a.relu_y + off);
        rsv[u] = a.rowscale != nullptr ? a.rowscale[min(r0 + rr, row_last)] : 1.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * kRowThreads;
        if (idx >= 64 * gq) break;
        const int rr = idx / gq, c4 = idx - rr * gq;
        const int o = 4 * c4;
        float v[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
        if (gbn) {
          const float yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float xh = (yy[s] - gv[NO + o + s]) * gv[2 * NO + o + s];
            v[s] = gv[o + s] * (v[s] - gv[3 * NO + o + s] - xh * gv[4 * NO + o + s]);
          }
        }
        if (a.relu_y != nullptr) {
          if (!(rv[u].x > 0.0f)) v[0] = 0.0f;
          if (!(rv[u].y > 0.0f)) v[1] = 0.0f;
          if (!(rv[u].z > 0.0f)) v[2] = 0.0f;
          if (!(rv[u].w > 0.0f)) v[3] = 0.0f;
        }
        const bool ok = r0 + rr < a.M;
        *reinterpret_cast<float4*>(gt + rr * GP + o) =
            make_float4(ok ? v[0] * rsv[u] : 0.0f, ok ? v[1] * rsv[u] : 0.0f, ok ? v[2] * rsv[u] : 0.0f,
                        ok ? v[3] * rsv[u] : 0.0f);
      }
    }
    __syncthreads();
    RL_STAMP_X(2);
    if (active) {
      for (int i = 0; i < per; ++i) {
        const int rt = rt0 + i;
        const int row = r0 + 16 * rt + lq;
        const bool rok = row < a.M;
        const int64_t off = (int64_t)min(row, row_last) * a.KI + k_base + 16 * t + 4 * g;
        const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float4 pv4 = z4, dv4 = z4, ay4 = z4, sy4 = z4;
        if (a.add_plain != nullptr) pv4 = *reinterpret_cast<const float4*>(a.add_plain + off);
        if (a.add_dout != nullptr) {
          dv4 = *reinterpret_cast<const float4*>(a.add_dout + off);
          ay4 = *reinterpret_cast<const float4*>(a.add_y + off);
        }
        const int ksum = k_base + 16 * t + 4 * g;
        if (want_sums && ksum < DS)
          sy4 = *reinterpret_cast<const float4*>(a.sum_y + (int64_t)min(row, row_last) * DS + ksum);
        // dX^T tile (k = k_base + 16t + 4g' + r, row): A[k = lq][o = 16j + 4g + s] = W[o][k]
        Feat<NO> gf;
        load_row<NO>(gf, gt + (16 * rt + lq) * GP, g);
        f32x4 acc = zero4();
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma16(wA[j][s], gf.f[j][s], acc);
        const int kl = 16 * t + 4 * g, k = k_base + kl;
        float v[4] = {acc[0] + pv4.x, acc[1] + pv4.y, acc[2] + pv4.z, acc[3] + pv4.w};
        if (a.add_dout != nullptr) {  // residual branch: BatchNorm backward of add_dout
          const float dd[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
          const float yy[4] = {ay4.x, ay4.y, ay4.z, ay4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xh = (yy[r] - ev[ks + kl + r]) * ev[2 * ks + kl + r];
            v[r] += ev[kl + r] * (dd[r] - ev[3 * ks + kl + r] - xh * ev[4 * ks + kl + r]);
          }
        }
        if (rok)
          *reinterpret_cast<float4*>(dx_at(a, row, k)) = make_float4(v[0], v[1], v[2], v[3]);
        if (want_sums) {  // sum(dx), sum(dx * xhat) over rows, for the BatchNorm that produced x
          const float yy[4] = {sy4.x, sy4.y, sy4.z, sy4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xh = (yy[r] - ev[5 * ks + kl + r]) * ev[6 * ks + kl + r];
            const float s1 = rok ? v[r] : 0.0f;
            sum1[r] += s1;        // lane-local over this lane's rows; reduced over the 16 lanes once
            sum2[r] += s1 * xh;
          }
        }
      }
    }
  }
  RL_STAMP_X(3);
  if (want_sums) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s1 = row16_sum(sum1[r]), s2 = row16_sum(sum2[r]);
      if (lq == 0) {
        red[(w * 2 + 0) * 16 + 4 * g + r] = active ? s1 : 0.0f;
        red[(w * 2 + 1) * 16 + 4 * g + r] = active ? s2 : 0.0f;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * ks; i += kRowThreads) {
      const int which = i / ks, kl = i - which * ks;
      if (kl < ntile * 16) {
        const int tile = kl >> 4, c = kl & 15;
        float s = 0.0f;
        for (int gi = 0; gi < groups; ++gi) s += red[((gi * ntile + tile) * 2 + which) * 16 + c];
        if (k_base + kl < DS) a.sum_out[((int64_t)rg * 2 + which) * DS + k_base + kl] = s;
      }
    }
  }
  RL_STAMP_X(4);
}

// dW/db role, LDS-staged: a workgroup = one row chunk x a group of 4 output tiles x one 64-column slice of x
// (one per wave).  The chunk's gradient slice g[64 rows][64 outputs] (all transforms applied) and
// x[64 rows][KI] (seen through its BatchNorm) are staged by all 256 threads with 16-byte loads -
// one memory latency for the whole tile instead of one per 16-row block - and the MFMA operands
// come from LDS (pitch = 16 mod 32 floats: conflict-free column reads).
template <int KI>
__device__ void rowlin_dw_role_lds(const RowLinArgs& a, const RowLinGeom& ge, const float* gv,
                                   float* lds_free, int lq, int g) {
  // a workgroup also takes ONE 64-column slice of x (KS columns): wide inputs are split over more
  // workgroups instead of lengthening the staging and the MFMA chain of each (the dW role is the long
  // pole of the launch for KI = 128)
  constexpr int KS = KI > 64 ? 64 : KI, NKS = KI / KS;
  constexpr int KT = KS / 16;
  constexpr int GP = 64 + 16, XP = KS + 16;
  const int not_ = a.NO / 16;
  const int n_og = (not_ + kRowWaves - 1) / kRowWaves;
  const int bi = blockIdx.x - ge.dx_blocks;
  const int ksi = bi % NKS, bj = bi / NKS;
  const int k0 = ksi * KS;
  const int og = bj % n_og, rc = bj / n_og;
  const int o_base = og * 64;
  const int ow = min(64, a.NO - o_base);  // outputs of this group
  float* gt = lds_free;                   // [64][GP]
  float* xt = gt + 64 * GP;               // [64][XP]
  const int DSW = a.x2 != nullptr ? a.x_split : KI;
  const int nrb16 = (a.M + 15) / 16;
  const int per = (nrb16 + ge.RC - 1) / ge.RC;  // 16-row blocks per chunk
  const int row_lo = rc * per * 16, row_hi = min((rc + 1) * per * 16, a.M);
  const int ot = og * kRowWaves + wave_id();
  const bool wave_on = ot < not_;
  const bool gbn = a.g_y != nullptr;
  f32x4 acc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) acc[kt] = zero4();
  float db = 0.0f;
  for (int r0 = row_lo; r0 < row_hi; r0 += 64) {
    if (r0 > row_lo) __syncthreads();
    // stage g slice (64 rows x ow / 4 float4) and x (64 rows x KI / 4 float4, seen through its
    // BatchNorm): four items per thread per pass, every load of the pass (clamped row, unconditional)
    // issued before the first use - one memory latency per pass instead of one per item
    const int gq = ow / 4;
    const int row_last = row_hi - 1;
    for (int base = threadIdx.x; base < 64 * gq; base += 4 * kRowThreads) {
      float4 dv[4], yv[4], rv[4];
      float rsv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = min(base + u * kRowThreads, 64 * gq - 1);
        const int rr = idx / gq, c4 = idx - rr * gq;
        const int64_t off = (int64_t)min(r0 + rr, row_last) * a.NO + o_base + 4 * c4;
        dv[u] = *reinterpret_cast<const float4*>(a.dy + off);
        if (gbn) yv[u] = *reinterpret_cast<const float4*>(a.g_y + off);
        if (a.relu_y != nullptr) rv[u] = *reinterpret_cast<const float4*>(a.relu_y + off);
        rsv[u] = a.rowscale != nullptr ? a.rowscale[min(r0 + rr, row_last)] : 1.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * kRowThreads;
        if (idx >= 64 * gq) break;
        const int rr = idx / gq, c4 = idx - rr * gq;
        const int o = o_base + 4 * c4;
        float v[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
        if (gbn) {
          const float yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float xh = (yy[s] - gv[a.NO + o + s]) * gv[2 * a.NO + o + s];
            v[s] = gv[o + s] * (v[s] - gv[3 * a.NO + o + s] - xh * gv[4 * a.NO + o + s]);
          }
        }
        if (a.relu_y != nullptr) {
          if (!(rv[u].x > 0.0f)) v[0] = 0.0f;
          if (!(rv[u].y > 0.0f)) v[1] = 0.0f;
          if (!(rv[u].z > 0.0f)) v[2] = 0.0f;
          if (!(rv[u].w > 0.0f)) v[3] = 0.0f;
        }
        const bool ok = r0 + rr < row_hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) gt[rr * GP + 4 * c4 + s] = ok ? v[s] * rsv[u] : 0.0f;
      }
    }
    for (int base = threadIdx.x; base < 64 * (KS / 4); base += 4 * kRowThreads) {
      float4 xv[4], sc[4], sh[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = min(base + u * kRowThreads, 64 * (KS / 4) - 1);
        const int rr = idx / (KS / 4), k = 4 * (idx - rr * (KS / 4));
        xv[u] = *reinterpret_cast<const float4*>(x_at(a, min(r0 + rr, row_last), k0 + k));
        if (a.x_bn != nullptr) {   // (concatenated operand: the parameter block covers the x part, DSW columns)
          const int kb = k0 + k < DSW ? k0 + k : 0;
          sc[u] = *reinterpret_cast<const float4*>(a.x_bn + kb);
          sh[u] = *reinterpret_cast<const float4*>(a.x_bn + DSW + kb);
          if (k0 + k >= DSW) {
            sc[u] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            sh[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * kRowThreads;
        if (idx >= 64 * (KS / 4)) break;
        const int rr = idx / (KS / 4), k = 4 * (idx - rr * (KS / 4));
        float v[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
        if (a.x_bn != nullptr) {
          v[0] = v[0] * sc[u].x + sh[u].x;
          v[1] = v[1] * sc[u].y + sh[u].y;
          v[2] = v[2] * sc[u].z + sh[u].z;
          v[3] = v[3] * sc[u].w + sh[u].w;
        }
        const bool ok = r0 + rr < row_hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) xt[rr * XP + k + s] = ok ? v[s] : 0.0f;
      }
    }
    RL_STAMP_W(2);
    __syncthreads();
    RL_STAMP_W(3);
    if (wave_on) {
      const int ol = 16 * wave_id() + lq;
#pragma unroll 4
      for (int st = 0; st < 16; ++st) {
        const int rr = 4 * st + g;
        const float gvv = gt[rr * GP + ol];
        db += gvv;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[kt] = mfma16(gvv, xt[rr * XP + 16 * kt + lq], acc[kt]);
      }
    }
  }
  RL_STAMP_W(4);
  if (!wave_on) return;
  float* p = a.partial + (int64_t)rc * (a.partial_ld > 0 ? (int64_t)a.partial_ld : (int64_t)a.NO * KI + a.NO);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) p[(int64_t)(16 * ot + 4 * g + r) * KI + k0 + 16 * kt + lq] = acc[kt][r];
  db += shfl_xor(db, 16);
  db += shfl_xor(db, 32);
  if (g == 0 && ksi == 0) p[(int64_t)a.NO * KI + 16 * ot + lq] = db;
  RL_STAMP_W(5);
}

template <int KI, int NO>
__global__ __launch_bounds__(kRowThreads) void rowlin_bwd_kernel(RowLinArgs a, RowLinGeom ge) {
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  RL_STAMP_X(0);
  RL_STAMP_W(0);
  float* gv = feta_lds;  // [5][NO] parameters of the gradient-side BatchNorm backward
  float* after = gv;
  if (a.g_y != nullptr) {
    float* scr = gv + 5 * NO;
    after = scr;
    // the BatchNorm parameter block is requested before the partial sums are reduced (one round trip less)
    const int cpre = min((int)threadIdx.x, NO - 1);
    const float bn_scale = a.g_bn[cpre], bn_mean = a.g_bn[2 * NO + cpre], bn_rstd = a.g_bn[3 * NO + cpre];
    if (a.g_sum != nullptr) {
      reduce_partials_t<kRowThreads, 32>(a.g_sum, a.Gs, NO, scr + 2 * NO, scr);
      for (int c = threadIdx.x; c < NO; c += kRowThreads) {
        gv[3 * NO + c] = scr[c] / (float)a.M;
        gv[4 * NO + c] = scr[NO + c] / (float)a.M;
        if (blockIdx.x == 0) {
          if (a.dbeta != nullptr) a.dbeta[c] = scr[c];
          if (a.dgamma != nullptr) a.dgamma[c] = scr[NO + c];
          if (a.g_fin_out != nullptr) {
            a.g_fin_out[c] = gv[3 * NO + c];
            a.g_fin_out[NO + c] = gv[4 * NO + c];
          }
        }
      }
    } else {
      for (int c = threadIdx.x; c < 2 * NO; c += kRowThreads) gv[3 * NO + c] = a.g_fin[c];
    }
    static_assert(NO <= kRowThreads, "one thread per column of the parameter block");
    if ((int)threadIdx.x < NO) {
      gv[threadIdx.x] = bn_scale;
      gv[NO + threadIdx.x] = bn_mean;
      gv[2 * NO + threadIdx.x] = bn_rstd;
    }
    __syncthreads();
  }
  RL_STAMP_W(1);
  if ((int)blockIdx.x < ge.dx_blocks) {
    rowlin_dx_role<NO>(a, ge, gv, after, lq, g);
  } else {
    rowlin_dw_role_lds<KI>(a, ge, gv, after, lq, g);
  }
}

// ---- BatchNorm1d (training mode) over M rows, stand-alone passes --------------------------------

struct BnArgs {
  const float* y;       // [M, D] pre-norm input
  const float* stats;   // [G, 2, D] partial (sum, sumsq)
  float* stats_out;     // same, written by bn_stats_kernel
  const float* gamma;
  const float* beta;
  const float* dout;    // [M, D]
  const float* mean_rstd_in;
  float* out;           // [M, D]
  float* mean_rstd;     // [2, D]
  const float* shift;   // bn_stats: [D] shift of the partial sums (feta_rowops.h) or null
  float* running_mean;  // [D] or null
  float* running_var;   // [D] or null
  int64_t* nbt;         // num_batches_tracked or null
  float* partial;       // [G, 2, D] backward partial sums
  float* dy;
  float* dgamma;
  float* dbeta;
  float momentum, eps;
  int M, D, G;
  int Gs;               // number of partial rows in `stats` (0: G)
  int prm4;             // mean_rstd_in is a [4][D] parameter block (mean at row 2, rstd at row 3)
};

// per-block partial (sum, sumsq) of y: threads = columns x row slices
__global__ __launch_bounds__(256) void bn_stats_kernel(BnArgs a) {
  float* red = feta_lds;  // [slices][2][D]
  const int D = a.D;
  const int slices = 256 / D > 0 ? 256 / D : 1;
  const int col = threadIdx.x % D, slice = threadIdx.x / D;
  const bool active = (int)threadIdx.x < slices * D;
  const int nrb = (a.M + kRowsPerBlock - 1) / kRowsPerBlock;
  float s1 = 0.0f, s2 = 0.0f;
  const float kshift = (active && a.shift != nullptr) ? a.shift[col] : 0.0f;   // shifted sums (feta_rowops.h)
  if (active) {
    for (int rb = blockIdx.x; rb < nrb; rb += a.G) {
      const int r0 = rb * kRowsPerBlock;
      for (int r = r0 + slice; r < min(r0 + kRowsPerBlock, a.M); r += slices) {
        const float v = a.y[(int64_t)r * D + col] - kshift;
        s1 += v;
        s2 += v * v;
      }
    }
    red[(slice * 2 + 0) * D + col] = s1;
    red[(slice * 2 + 1) * D + col] = s2;
  }
  __syncthreads();
  if ((int)threadIdx.x < D) {
    float t1 = 0.0f, t2 = 0.0f;
    for (int s = 0; s < slices; ++s) {
      t1 += red[(s * 2 + 0) * D + col];
      t2 += red[(s * 2 + 1) * D + col];
    }
    a.stats_out[((int64_t)blockIdx.x * 2 + 0) * D + col] = t1;
    a.stats_out[((int64_t)blockIdx.x * 2 + 1) * D + col] = t2;
    if (blockIdx.x == 0) a.stats_out[(int64_t)a.G * 2 * D + col] = kshift;   // row G: the shift
  }
}

// out = gamma (y - mean) rstd + beta; every block re-reduces the G partials (deterministic,
// no extra launch); block 0 publishes mean/rstd and updates the running statistics.
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(BnArgs a) {
  const int D = a.D;
  float* sc = feta_lds;            // [D] scale = gamma * rstd
  float* sh = feta_lds + D;        // [D] shift = beta - mean * scale
  float* tot = feta_lds + 2 * D;   // [2][D]
  float* red = feta_lds + 4 * D;   // [slices][2][D]
  reduce_partials(a.stats, a.Gs > 0 ? a.Gs : a.G, D, red, tot);
  for (int c = threadIdx.x; c < D; c += 256) {
    float mean, var;
    bn_moments(a.stats, a.Gs > 0 ? a.Gs : a.G, D, a.M, tot, c, mean, var);
    const float rstd = rsqrtf(var + a.eps);
    const float scale = a.gamma[c] * rstd;
    sc[c] = scale;
    sh[c] = a.beta[c] - mean * scale;
    if (blockIdx.x == 0) {
      if (a.prm4) {
        a.mean_rstd[c] = scale;
        a.mean_rstd[D + c] = sh[c];
        a.mean_rstd[2 * D + c] = mean;
        a.mean_rstd[3 * D + c] = rstd;
      } else {
        a.mean_rstd[c] = mean;
        a.mean_rstd[D + c] = rstd;
      }
      if (a.running_mean != nullptr) {
        const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
        a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * mean;
        a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * unbiased;
      }
      if (c == 0 && a.nbt != nullptr) *a.nbt += 1;
    }
  }
  __syncthreads();
  const int64_t n4 = (int64_t)a.M * D / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i * 4) % D);
    const float4 v = reinterpret_cast<const float4*>(a.y)[i];
    reinterpret_cast<float4*>(a.out)[i] =
        make_float4(v.x * sc[c] + sh[c], v.y * sc[c + 1] + sh[c + 1], v.z * sc[c + 2] + sh[c + 2],
                    v.w * sc[c + 3] + sh[c + 3]);
  }
}

// backward pass 1: per-block partial sums of dout and dout * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnArgs a) {
  float* red = feta_lds;
  const int D = a.D;
  const int slices = 256 / D > 0 ? 256 / D : 1;
  const int col = threadIdx.x % D, slice = threadIdx.x / D;
  const bool active = (int)threadIdx.x < slices * D;
  const int nrb = (a.M + kRowsPerBlock - 1) / kRowsPerBlock;
  const int mo = a.prm4 ? 2 * D : 0;
  float s1 = 0.0f, s2 = 0.0f;
  if (active) {
    const float mean = a.mean_rstd_in[mo + col], rstd = a.mean_rstd_in[mo + D + col];
    for (int rb = blockIdx.x; rb < nrb; rb += a.G) {
      const int r0 = rb * kRowsPerBlock;
      for (int r = r0 + slice; r < min(r0 + kRowsPerBlock, a.M); r += slices) {
        const float d = a.dout[(int64_t)r * D + col];
        s1 += d;
        s2 += d * (a.y[(int64_t)r * D + col] - mean) * rstd;
      }
    }
    red[(slice * 2 + 0) * D + col] = s1;
    red[(slice * 2 + 1) * D + col] = s2;
  }
  __syncthreads();
  if ((int)threadIdx.x < D) {
    float t1 = 0.0f, t2 = 0.0f;
    for (int s = 0; s < slices; ++s) {
      t1 += red[(s * 2 + 0) * D + col];
      t2 += red[(s * 2 + 1) * D + col];
    }
    a.partial[((int64_t)blockIdx.x * 2 + 0) * D + col] = t1;
    a.partial[((int64_t)blockIdx.x * 2 + 1) * D + col] = t2;
  }
}

// backward pass 2: dy = gamma rstd (dout - mean(dout) - xhat mean(dout xhat)); block 0 writes
// dgamma = sum(dout xhat), dbeta = sum(dout)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnArgs a) {
  const int D = a.D;
  float* k0 = feta_lds;           // gamma*rstd
  float* mu = feta_lds + D;
  float* rs = feta_lds + 2 * D;
  float* tot = feta_lds + 3 * D;  // [2][D]: sum(dout), sum(dout*xhat)
  float* red = feta_lds + 5 * D;
  reduce_partials(a.partial, a.G, D, red, tot);
  for (int c = threadIdx.x; c < D; c += 256) {
    const float rstd = a.mean_rstd_in[D + c];
    k0[c] = a.gamma[c] * rstd;
    mu[c] = a.mean_rstd_in[c];
    rs[c] = rstd;
    if (blockIdx.x == 0) {
      a.dbeta[c] = tot[c];
      a.dgamma[c] = tot[D + c];
    }
  }
  __syncthreads();
  const float inv_m = 1.0f / (float)a.M;
  const int64_t n4 = (int64_t)a.M * D / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i * 4) % D);
    const float4 d = reinterpret_cast<const float4*>(a.dout)[i];
    const float4 v = reinterpret_cast<const float4*>(a.y)[i];
    float o[4];
    const float dd[4] = {d.x, d.y, d.z, d.w};
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float xh = (vv[t] - mu[c + t]) * rs[c + t];
      o[t] = k0[c + t] * (dd[t] - tot[c + t] * inv_m - xh * tot[D + c + t] * inv_m);
    }
    reinterpret_cast<float4*>(a.dy)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---- host side --------------------------------------------------------------------------------

int row_blocks(int M) {
  const int nrb = (M + kRowsPerBlock - 1) / kRowsPerBlock;
  return nrb < kMaxRowBlocks ? nrb : kMaxRowBlocks;
}
int row_chunks(int M) {
  const int nrb16 = (M + 15) / 16;
  const int rc = (nrb16 + kChunkRowBlocks - 1) / kChunkRowBlocks;
  return rc < 1 ? 1 : (rc > kMaxChunks ? kMaxChunks : rc);
}

bool dim_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128 || c == 192 || c == 256; }

// tiles per workgroup such that the staged weight slice fits beside the other LDS users
// tiles per workgroup (tuning knobs FETA_ROWLIN_TG / FETA_ROWLIN_TG_DX = 1..4).  Measured on the
// BASELINE batch (M = 4736 rows): 2 output tiles per workgroup beat 4 in the forward (more, shorter
// workgroups: 0.520 -> 0.488 ms per step); see DESIGN.md section 6.
int tg_env(const char* name, int dflt) {
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  return (v >= 1 && v <= 4) ? v : dflt;
}
// Output tiles per workgroup of the forward kernel.  Its ~230 registers allow two workgroups per CU - 512 at a time on the
// MI355X - so a grid just above that runs a second, mostly empty round: in_proj at config 4 (M = 8192, 192 outputs) was
// 128 row blocks x 6 groups of two tiles = 768 workgroups, 15 us; as 4 groups of three tiles it is 512, one round (PATTERN
// B = 64, N_pad = 128: 0.4216 -> 0.4115 ms per step).  Two tiles otherwise (more workgroups in flight for small M).
int tiles_fwd(int KI, int NO, int G) {  // LDS: 16 tg (KI + 4) floats
  const int n_ot = NO / 16;
  int tg = tg_env("FETA_ROWLIN_TG", 0);
  if (tg <= 0) {
    // the fewest tiles per workgroup among {2, 3, 4} that reach the fewest whole rounds (N_pad = 188, M = 12032: 1128
    // workgroups = 3 rounds with two tiles, 752 = 2 rounds with three)
    tg = 2;
    int best = (G * ((n_ot + 1) / 2) + 511) / 512;
    for (int cand = 3; cand <= 4; ++cand) {
      const int rounds = (G * ((n_ot + cand - 1) / cand) + 511) / 512;
      if (rounds < best) {
        best = rounds;
        tg = cand;
      }
    }
  }
  while (tg > 1 && 16 * tg * (KI + 4) * 4 > 48 * 1024) --tg;
  return tg;
}
int tiles_dx(int NO) {   // LDS: NO (16 tg + 4) floats (two workgroups per CU: up to 64 KB each - four tiles at NO = 192, where
  // three made the in_proj backward 256 + 384 = 640 workgroups, a second round; four: 128 + 384 = 512)
  int tg = tg_env("FETA_ROWLIN_TG_DX", 4);
  while (tg > 1 && NO * (16 * tg + 4) * 4 > 64 * 1024) --tg;
  return tg;
}

#define FETA_DIM_SWITCH(VAL, CALL)  \
  switch (VAL) {                    \
    case 16: CALL(16); break;       \
    case 32: CALL(32); break;       \
    case 64: CALL(64); break;       \
    case 128: CALL(128); break;     \
    case 192: CALL(192); break;     \
    default: CALL(256); break;      \
  }

template <int KI>
void launch_bwd_ki(const RowLinArgs& a, const RowLinGeom& ge, int grid, size_t lds, hipStream_t stream) {
#define CALL(NOV)                                                                   \
  {                                                                                 \
    auto kern = rowlin_bwd_kernel<KI, NOV>;                                         \
    static LdsSeen lds_seen;                                                     \
    allow_dynamic_lds(kern, lds, lds_seen);                                         \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kRowThreads), lds, stream, a, ge);    \
  }
  FETA_DIM_SWITCH(a.NO, CALL)
#undef CALL
}

}  // namespace feta

using namespace feta;

extern "C" int feta_rowlin_blocks(int M) { return row_blocks(M); }
extern "C" int feta_rowlin_chunks(int M) { return row_chunks(M); }

extern "C" int feta_rowlin_fwd_ex(const feta_rowlin_ex* d, feta_stream_t stream) {
  FETA_REQUIRE(d != nullptr, "rowlin_fwd_ex: null descriptor");
  const RowLinArgs& a = *d;
  FETA_REQUIRE(a.x && a.w && a.y && a.M > 0, "rowlin_fwd: null pointer / empty");
  FETA_REQUIRE(dim_ok(a.KI) && (a.NO % 16) == 0 && a.NO > 0 && a.NO <= 1024,
               "rowlin_fwd: unsupported dims KI=%d NO=%d", a.KI, a.NO);
  FETA_REQUIRE(aligned16(a.x) && aligned16(a.w) && aligned16(a.y) && (!a.residual || aligned16(a.residual)),
               "rowlin_fwd: pointers must be 16-byte aligned");
  FETA_REQUIRE(!a.x_stats || (a.x_gamma && a.x_beta && a.x_bn_out && a.Gx > 0),
               "rowlin_fwd: x_stats needs gamma, beta, bn_out, Gx");
  FETA_REQUIRE(!a.res_bn || a.residual, "rowlin_fwd: res_bn without residual");
  FETA_REQUIRE(!a.x2 || (a.x_split > 0 && a.x_split < a.KI && (a.x_split % 16) == 0 && aligned16(a.x2)),
               "rowlin_fwd: x2 needs 0 < x_split < KI, x_split %% 16 == 0 (an input BatchNorm covers the x part)");
  RowLinGeom ge{};
  ge.G = row_blocks(a.M);
  ge.TG = tiles_fwd(a.KI, a.NO, ge.G);
  const int n_og = (a.NO / 16 + ge.TG - 1) / ge.TG;
  const size_t lds = sizeof(float) * (16 * ge.TG * (a.KI + 4) + kRowWaves * 2 * 16 * ge.TG + 2 * a.KI +
                                      (a.x_stats ? reduce_scratch_floats(a.x2 ? a.x_split : a.KI) : 0));
#define CALL(KV)                                                                                     \
  {                                                                                                  \
    auto kern = rowlin_fwd_kernel<KV>;                                                               \
    hipLaunchKernelGGL(kern, dim3(ge.G * n_og), dim3(kRowThreads), lds, (hipStream_t)stream, a, ge); \
  }
  FETA_DIM_SWITCH(a.KI, CALL)
#undef CALL
  return check_launch("feta_rowlin_fwd");
}

#ifdef FETA_TIMING
extern "C" int feta_debug_rowlin_stamps(unsigned long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(feta_rowlin_stamps), sizeof(unsigned long long) * 32);
}
#endif

extern "C" int feta_rowlin_bwd_ex(const feta_rowlin_ex* d, float* dwdb, feta_stream_t stream) {
  FETA_REQUIRE(d != nullptr, "rowlin_bwd_ex: null descriptor");
  const RowLinArgs& a = *d;
  // dx == NULL: weight / bias gradient only (the dX chain ran elsewhere)
  FETA_REQUIRE(a.x && a.w && a.dy && a.partial && a.M > 0, "rowlin_bwd: null pointer / empty");
  FETA_REQUIRE(a.dx || (!a.add_plain && !a.add_dout && !a.sum_out && !a.dx2), "rowlin_bwd: dW-only launch with dX epilogue operands");
  FETA_REQUIRE(dwdb || a.partial_ld > 0, "rowlin_bwd: dwdb may only be NULL with a caller-reduced partial_ld");
  FETA_REQUIRE(dim_ok(a.KI) && dim_ok(a.NO), "rowlin_bwd: unsupported dims KI=%d NO=%d", a.KI, a.NO);
  FETA_REQUIRE(aligned16(a.x) && aligned16(a.w) && aligned16(a.dy) && aligned16(a.dx) &&  /* NULL is aligned */
                   (!a.relu_y || aligned16(a.relu_y)) && (!a.g_y || aligned16(a.g_y)),
               "rowlin_bwd: pointers must be 16-byte aligned");
  FETA_REQUIRE(!a.g_y || (a.g_bn && (a.g_sum || a.g_fin)), "rowlin_bwd: g_y needs g_bn and g_sum|g_fin");
  FETA_REQUIRE(!a.g_sum || a.Gs > 0, "rowlin_bwd: g_sum needs Gs");
  FETA_REQUIRE(!a.add_dout || (a.add_y && a.add_bn && a.add_fin), "rowlin_bwd: incomplete add_* set");
  FETA_REQUIRE(!a.sum_out || (a.sum_y && a.sum_bn), "rowlin_bwd: incomplete sum_* set");
  FETA_REQUIRE(!a.x2 == !a.dx2, "rowlin_bwd: x2 and dx2 go together");
  FETA_REQUIRE(!a.x2 || (a.x_split > 0 && a.x_split < a.KI && (a.x_split % 16) == 0 && aligned16(a.x2) &&
                         aligned16(a.dx2) && !a.add_plain && !a.add_dout),
               "rowlin_bwd: x2 needs 0 < x_split < KI, x_split %% 16 == 0 and no add_* (x_bn / sum_* cover the x part)");
  RowLinGeom ge{};
  ge.RC = row_chunks(a.M);
  ge.G = row_blocks(a.M);
  ge.TG = tiles_dx(a.NO);
  const int n_kg = (a.KI / 16 + ge.TG - 1) / ge.TG;
  ge.dx_blocks = a.dx != nullptr ? ge.G * n_kg : 0;
  const int n_ot = a.NO / 16;
  const int n_ks = a.KI > 64 ? a.KI / 64 : 1;   // 64-column slices of x, one per dW workgroup
  const int dw_blocks = ge.RC * ((n_ot + kRowWaves - 1) / kRowWaves) * n_ks;
  int grid = ge.dx_blocks + dw_blocks;
  if (const char* role = getenv("FETA_ROWLIN_ROLE")) {  // diagnostic: time one role alone (results incomplete)
    if (role[0] == 'x') grid = ge.dx_blocks;
    if (role[0] == 'w') { ge.dx_blocks = 0; grid = dw_blocks; }
  }
  const size_t dx_lds = 64 * (a.NO + 4) + 7 * 16 * ge.TG + kRowWaves * 2 * 16;
  const size_t dw_lds = 64 * (64 + 16) + 64 * ((a.KI > 64 ? 64 : a.KI) + 16);
  const size_t lds = sizeof(float) * ((dx_lds > dw_lds ? dx_lds : dw_lds) +
                                      (a.g_y ? 5 * a.NO + reduce_scratch_floats(a.NO) : 0));
#define CALL(KV) launch_bwd_ki<KV>(a, ge, grid, lds, (hipStream_t)stream);
  FETA_DIM_SWITCH(a.KI, CALL)
#undef CALL
  int rc = check_launch("feta_rowlin_bwd");
  if (rc != FETA_OK || dwdb == nullptr) return rc;  // NULL dwdb: the caller reduces all partials at once
  FETA_REQUIRE(a.partial_ld == 0, "rowlin_bwd: dwdb with partial_ld is not supported");
  return feta_colsum(a.partial, dwdb, ge.RC, a.NO * a.KI + a.NO, stream);
}

extern "C" int feta_rowlin_fwd(const float* x, const float* w, const float* bias,
                               const float* rowscale, const float* residual, float* y, float* stats,
                               int relu, int M, int KI, int NO, feta_stream_t stream) {
  feta_rowlin_ex a{};
  a.x = x; a.w = w; a.bias = bias; a.rowscale = rowscale; a.residual = residual; a.y = y;
  a.stats = stats; a.relu = relu; a.M = M; a.KI = KI; a.NO = NO;
  return feta_rowlin_fwd_ex(&a, stream);
}

extern "C" int feta_rowlin_bwd(const float* x, const float* w, const float* dy,
                               const float* rowscale, const float* ysaved, float* dx, float* partial,
                               float* dwdb, int M, int KI, int NO, feta_stream_t stream) {
  feta_rowlin_ex a{};
  a.x = x; a.w = w; a.dy = dy; a.rowscale = rowscale; a.relu_y = ysaved; a.dx = dx;
  a.partial = partial; a.M = M; a.KI = KI; a.NO = NO;
  return feta_rowlin_bwd_ex(&a, dwdb, stream);
}

extern "C" int feta_bn_stats(const float* y, float* stats, int M, int D, feta_stream_t stream) {
  return feta_bn_stats_shift(y, nullptr, stats, M, D, stream);
}

extern "C" int feta_bn_stats_shift(const float* y, const float* shift, float* stats, int M, int D, feta_stream_t stream) {
  FETA_REQUIRE(y && stats && M > 0 && D > 0 && D <= 256, "bn_stats: need 0 < D <= 256");
  BnArgs a{};
  a.y = y; a.shift = shift; a.stats_out = stats; a.M = M; a.D = D; a.G = row_blocks(M);
  const int slices = 256 / D > 0 ? 256 / D : 1;
  auto kern = bn_stats_kernel;
  hipLaunchKernelGGL(kern, dim3(a.G), dim3(256), sizeof(float) * slices * 2 * D, (hipStream_t)stream, a);
  return check_launch("feta_bn_stats");
}

static int bn_apply_launch(BnArgs& a, feta_stream_t stream) {
  const int64_t n4 = (int64_t)a.M * a.D / 4;
  int grid = (int)((n4 + 255) / 256);
  grid = grid > 1024 ? 1024 : grid;
  const int slices = 256 / a.D > 0 ? 256 / a.D : 1;
  auto kern = bn_apply_fwd_kernel;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), sizeof(float) * (4 * a.D + reduce_red_floats(a.D)),
                     (hipStream_t)stream, a);
  return check_launch("feta_bn_apply_fwd");
}

extern "C" int feta_bn_apply_fwd(const float* y, const float* stats, const float* gamma,
                                 const float* beta, float* out, float* mean_rstd, float* running_mean,
                                 float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                 int M, int D, feta_stream_t stream) {
  FETA_REQUIRE(y && stats && gamma && beta && out && mean_rstd, "bn_apply_fwd: null pointer");
  FETA_REQUIRE(M > 0 && D > 0 && D <= 256 && (D % 4) == 0, "bn_apply_fwd: need D %% 4 == 0, D <= 256");
  FETA_REQUIRE(aligned16(y) && aligned16(out), "bn_apply_fwd: pointers must be 16-byte aligned");
  BnArgs a{};
  a.y = y; a.stats = stats; a.gamma = gamma; a.beta = beta; a.out = out; a.mean_rstd = mean_rstd;
  a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum; a.eps = eps;
  a.nbt = num_batches_tracked;
  a.M = M; a.D = D; a.G = row_blocks(M);
  return bn_apply_launch(a, stream);
}

extern "C" int feta_bn_apply_fwd_prm(const float* y, const float* stats, const float* gamma,
                                     const float* beta, float* out, float* bn_prm, float* running_mean,
                                     float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                     int M, int D, int G_stats, feta_stream_t stream) {
  FETA_REQUIRE(y && stats && gamma && beta && out && bn_prm, "bn_apply_fwd_prm: null pointer");
  FETA_REQUIRE(M > 0 && D > 0 && D <= 256 && (D % 4) == 0, "bn_apply_fwd_prm: need D %% 4 == 0, D <= 256");
  FETA_REQUIRE(aligned16(y) && aligned16(out), "bn_apply_fwd_prm: pointers must be 16-byte aligned");
  BnArgs a{};
  a.y = y; a.stats = stats; a.gamma = gamma; a.beta = beta; a.out = out; a.mean_rstd = bn_prm;
  a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum; a.eps = eps;
  a.nbt = num_batches_tracked;
  a.M = M; a.D = D; a.G = row_blocks(M); a.Gs = G_stats; a.prm4 = 1;
  return bn_apply_launch(a, stream);
}

extern "C" int feta_bn_bwd_reduce(const float* y, const float* dout, const float* bn_prm,
                                  float* partial, int M, int D, feta_stream_t stream) {
  FETA_REQUIRE(y && dout && bn_prm && partial && M > 0 && D > 0 && D <= 256, "bn_bwd_reduce: bad arguments");
  BnArgs a{};
  a.y = y; a.dout = dout; a.mean_rstd_in = bn_prm; a.partial = partial; a.M = M; a.D = D;
  a.G = row_blocks(M); a.prm4 = 1;
  const int slices = 256 / D > 0 ? 256 / D : 1;
  auto k1 = bn_bwd_reduce_kernel;
  hipLaunchKernelGGL(k1, dim3(a.G), dim3(256), sizeof(float) * slices * 2 * D, (hipStream_t)stream, a);
  return check_launch("feta_bn_bwd_reduce");
}

extern "C" int feta_bn_bwd(const float* y, const float* dout, const float* mean_rstd,
                           const float* gamma, float* partial, float* dy, float* dgamma, float* dbeta,
                           int M, int D, feta_stream_t stream) {
  FETA_REQUIRE(y && dout && mean_rstd && gamma && partial && dy && dgamma && dbeta, "bn_bwd: null pointer");
  FETA_REQUIRE(M > 0 && D > 0 && D <= 256 && (D % 4) == 0, "bn_bwd: need D %% 4 == 0, D <= 256");
  FETA_REQUIRE(aligned16(y) && aligned16(dout) && aligned16(dy), "bn_bwd: pointers must be 16-byte aligned");
  BnArgs a{};
  a.y = y; a.dout = dout; a.mean_rstd_in = mean_rstd; a.gamma = gamma; a.partial = partial; a.dy = dy;
  a.dgamma = dgamma; a.dbeta = dbeta; a.M = M; a.D = D; a.G = row_blocks(M);
  const int slices = 256 / D > 0 ? 256 / D : 1;
  auto k1 = bn_bwd_reduce_kernel;
  hipLaunchKernelGGL(k1, dim3(a.G), dim3(256), sizeof(float) * slices * 2 * D, (hipStream_t)stream, a);
  const int64_t n4 = (int64_t)M * D / 4;
  int grid = (int)((n4 + 255) / 256);
  grid = grid > 1024 ? 1024 : grid;
  auto k2 = bn_bwd_apply_kernel;
  hipLaunchKernelGGL(k2, dim3(grid), dim3(256), sizeof(float) * (5 * D + reduce_red_floats(D)),
                     (hipStream_t)stream, a);
  return check_launch("feta_bn_bwd");
}
