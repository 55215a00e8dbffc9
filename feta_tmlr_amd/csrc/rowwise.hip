// Row-wise (per-node) part of the encoder layer (A1) and the combine (A4): the in/out
// projections, the FFN linears and linear_cat with their element-wise neighbours fused, and
// training-mode BatchNorm1d over all N*B rows.  Replaces, per DiffTransformerEncoderLayer
// call (contract transformer/models.py:166-167; body per upstream GraphiT, README.md:129):
//   F.linear(+bias) -> [relu] -> [* degree] -> [+ residual]      one launch, plus the BN
//   statistics of the result as per-block partial sums (no separate statistics pass);
//   backward: dX and the split-K weight/bias gradient in ONE launch (blocks take roles), the
//   partials reduced deterministically by colsum_kernel.
// Activations are [M, C] row-major with M = N*B rows (seq-first rows are contiguous).
//
// Decomposition: a workgroup = 64 rows (one 16-row block per wave) x a group of up to 4
// output tiles; the weight slice of the group is staged ONCE in LDS by all 256 threads
// (padded pitch, conflict-free operand reads) - the small problem is latency-bound, so the
// dependent global loads of the weights are what has to go.
// MFMA orientation: Y^T tile = W_tile . X_tile^T, so the accumulator holds 4 consecutive
// output features of one row per lane -> 16-byte epilogue loads/stores, and the row-wise
// epilogue operands (degree, residual) are lane-local.
#include "feta_abi_common.h"
#include "feta_tiles.h"

namespace feta {

constexpr int kRowWaves = 4;
constexpr int kRowThreads = 64 * kRowWaves;
constexpr int kRowsPerBlock = 16 * kRowWaves;
constexpr int kMaxRowBlocks = 256;  // cap on per-block partial statistics
constexpr int kMaxChunks = 128;     // cap on split-K chunks of the weight gradient
constexpr int kChunkRowBlocks = 4;  // 16-row blocks per split-K chunk

struct RowLinArgs {
  const float* x;         // [M, KI]
  const float* w;         // [NO, KI]
  const float* bias;      // [NO] or null
  const float* rowscale;  // [M] or null
  const float* residual;  // [M, NO] or null
  const float* dy;        // [M, NO]              (backward)
  const float* ysaved;    // [M, NO] relu output  (backward, null = no relu)
  float* y;               // [M, NO]
  float* stats;           // [G, 2, NO] or null
  float* dx;              // [M, KI]
  float* partial;         // [RC, NO*KI + NO]
  int M, KI, NO, relu;
  int G;          // row groups of the grid (row blocks are strided over it)
  int TG;         // output tiles per workgroup (forward) / k tiles per workgroup (dX role)
  int RC;         // weight-gradient row chunks
  int dx_blocks;  // backward: blocks [0, dx_blocks) compute dX, the rest dW/db partials
};

// ---- forward ------------------------------------------------------------------------------
template <int KI>
__global__ __launch_bounds__(kRowThreads) void rowlin_fwd_kernel(RowLinArgs a) {
  constexpr int LDW = KI + 4;  // LDS pitch of a weight row (16-byte aligned, bank-shifted)
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int rg = blockIdx.x % a.G, og = blockIdx.x / a.G;
  const int tgw = a.TG * 16;
  const int o_base = og * tgw;
  const int ntile = min(a.TG, a.NO / 16 - og * a.TG);
  float* wt = feta_lds;               // [TG*16][LDW]
  float* red = feta_lds + tgw * LDW;  // [kRowWaves][2][TG*16]
  const bool want_stats = a.stats != nullptr;

  const int nvec = ntile * 16 * (KI / 4);
  for (int idx = threadIdx.x; idx < nvec; idx += kRowThreads) {
    const int row = idx / (KI / 4), c4 = idx - row * (KI / 4);
    *reinterpret_cast<float4*>(wt + row * LDW + 4 * c4) =
        *reinterpret_cast<const float4*>(a.w + (int64_t)(o_base + row) * KI + 4 * c4);
  }
  if (want_stats)
    for (int i = threadIdx.x; i < kRowWaves * 2 * tgw; i += kRowThreads) red[i] = 0.0f;
  __syncthreads();

  float* my = red + wave_id() * 2 * tgw;
  const int nrb = (a.M + kRowsPerBlock - 1) / kRowsPerBlock;
  for (int rb = rg; rb < nrb; rb += a.G) {
    const int row = rb * kRowsPerBlock + wave_id() * 16 + lq;
    const bool rok = row < a.M;
    Feat<KI> xf;
    load_row<KI>(xf, rok ? a.x + (int64_t)row * KI : nullptr, g);
    const float rs = (a.rowscale != nullptr && rok) ? a.rowscale[row] : 1.0f;
    for (int t = 0; t < ntile; ++t) {
      Feat<KI> wf;
      load_row<KI>(wf, wt + (16 * t + lq) * LDW, g);
      f32x4 acc = dot_rows<KI>(wf, xf, zero4());  // (o = o_base + 16t + 4g + r, row)
      const int ol = 16 * t + 4 * g, o = o_base + ol;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[r] + (a.bias != nullptr ? a.bias[o + r] : 0.0f);
        if (a.relu) v[r] = fmaxf(v[r], 0.0f);
        v[r] *= rs;
      }
      if (rok) {
        if (a.residual != nullptr) {
          const float4 rv = *reinterpret_cast<const float4*>(a.residual + (int64_t)row * a.NO + o);
          v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
        *reinterpret_cast<float4*>(a.y + (int64_t)row * a.NO + o) = make_float4(v[0], v[1], v[2], v[3]);
      }
      if (want_stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s1 = rok ? v[r] : 0.0f, s2 = s1 * s1;
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) {
            s1 += shfl_xor(s1, m);
            s2 += shfl_xor(s2, m);
          }
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
      }
    }
  }
}

// ---- backward: dX role (template on the contraction dim NO) and dW/db role (template on KI) ---

template <int NO>
__device__ void rowlin_dx_role(const RowLinArgs& a, int lq, int g) {
  const int rg = blockIdx.x % a.G, kg = blockIdx.x / a.G;
  const int ks = a.TG * 16, ldw = ks + 4;
  const int k_base = kg * ks;
  const int ntile = min(a.TG, a.KI / 16 - kg * a.TG);
  float* wt = feta_lds;  // [NO][ldw]: W[:, k_base : k_base + 16 ntile]
  const int rowvec = ntile * 4;
  for (int idx = threadIdx.x; idx < NO * rowvec; idx += kRowThreads) {
    const int o = idx / rowvec, c4 = idx - o * rowvec;
    *reinterpret_cast<float4*>(wt + o * ldw + 4 * c4) =
        *reinterpret_cast<const float4*>(a.w + (int64_t)o * a.KI + k_base + 4 * c4);
  }
  __syncthreads();
  const int nrb = (a.M + kRowsPerBlock - 1) / kRowsPerBlock;
  for (int rb = rg; rb < nrb; rb += a.G) {
    const int row = rb * kRowsPerBlock + wave_id() * 16 + lq;
    const bool rok = row < a.M;
    const float rs = (a.rowscale != nullptr && rok) ? a.rowscale[row] : 1.0f;
    Feat<NO> gf;  // g = dy * rowscale * [ysaved > 0]
    load_row<NO>(gf, rok ? a.dy + (int64_t)row * NO : nullptr, g, rs);
    if (a.ysaved != nullptr && rok) {
#pragma unroll
      for (int j = 0; j < Feat<NO>::NJ; ++j) {
        const int o = 16 * j + 4 * g;
        if (o < NO) {
          const float4 yv = *reinterpret_cast<const float4*>(a.ysaved + (int64_t)row * NO + o);
          if (!(yv.x > 0.0f)) gf.f[j][0] = 0.0f;
          if (!(yv.y > 0.0f)) gf.f[j][1] = 0.0f;
          if (!(yv.z > 0.0f)) gf.f[j][2] = 0.0f;
          if (!(yv.w > 0.0f)) gf.f[j][3] = 0.0f;
        }
      }
    }
    for (int t = 0; t < ntile; ++t) {
      // dX^T tile (k = k_base + 16t + 4g' + r, row): A[k = lq][o = 16j + 4g + s] = W[o][k]
      f32x4 acc = zero4();
#pragma unroll
      for (int j = 0; j < Feat<NO>::NJ; ++j) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int o = 16 * j + 4 * g + s;
          const float wv = o < NO ? wt[o * ldw + 16 * t + lq] : 0.0f;
          acc = mfma16(wv, gf.f[j][s], acc);
        }
      }
      if (rok)
        *reinterpret_cast<float4*>(a.dx + (int64_t)row * a.KI + k_base + 16 * t + 4 * g) =
            make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

template <int KI>
__device__ void rowlin_dw_role(const RowLinArgs& a, int lq, int g) {
  constexpr int KT = KI / 16;
  const int not_ = a.NO / 16;
  const int item = (blockIdx.x - a.dx_blocks) * kRowWaves + wave_id();
  if (item >= a.RC * not_) return;
  const int ot = item % not_, rc = item / not_;
  const int nrb16 = (a.M + 15) / 16;
  const int per = (nrb16 + a.RC - 1) / a.RC;
  const int o = 16 * ot + lq;
  f32x4 acc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) acc[kt] = zero4();
  float db = 0.0f;
  for (int rb = rc * per; rb < min((rc + 1) * per, nrb16); ++rb) {
    // issue every load of the row block before the first MFMA (one latency per block)
    float gv[4], xv[4][KT];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rb + 4 * g + r;
      const bool rok = row < a.M;
      float v = rok ? a.dy[(int64_t)row * a.NO + o] : 0.0f;
      if (rok && a.rowscale != nullptr) v *= a.rowscale[row];
      if (rok && a.ysaved != nullptr && !(a.ysaved[(int64_t)row * a.NO + o] > 0.0f)) v = 0.0f;
      gv[r] = v;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) xv[r][kt] = rok ? a.x[(int64_t)row * KI + 16 * kt + lq] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      db += gv[r];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc[kt] = mfma16(gv[r], xv[r][kt], acc[kt]);  // (o 4g+r', k lq)
    }
  }
  float* p = a.partial + (int64_t)rc * ((int64_t)a.NO * KI + a.NO);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) p[(int64_t)(16 * ot + 4 * g + r) * KI + 16 * kt + lq] = acc[kt][r];
  db += shfl_xor(db, 16);
  db += shfl_xor(db, 32);
  if (g == 0) p[(int64_t)a.NO * KI + o] = db;
}

template <int KI, int NO>
__global__ __launch_bounds__(kRowThreads) void rowlin_bwd_kernel(RowLinArgs a) {
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  if ((int)blockIdx.x < a.dx_blocks)
    rowlin_dx_role<NO>(a, lq, g);
  else
    rowlin_dw_role<KI>(a, lq, g);
}

// ---- BatchNorm1d (training mode) over M rows ---------------------------------------------

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
  float* running_mean;  // [D] or null
  float* running_var;   // [D] or null
  float* partial;       // [G, 2, D] backward partial sums
  float* dy;
  float* dgamma;
  float* dbeta;
  float momentum, eps;
  int M, D, G;
};

// sums the G partial pairs [G][2][D] with all 256 threads; on return tot[0][c], tot[1][c]
// (LDS, [2][D]) hold the totals.  red: [slices][2][D] scratch.
__device__ __forceinline__ void bn_reduce_partials(const float* part, int G, int D, float* red,
                                                   float* tot) {
  const int slices = 256 / D > 0 ? 256 / D : 1;
  const int col = threadIdx.x % D, slice = threadIdx.x / D;
  if ((int)threadIdx.x < slices * D) {
    float s1 = 0.0f, s2 = 0.0f;
    for (int gi = slice; gi < G; gi += slices) {
      s1 += part[((int64_t)gi * 2 + 0) * D + col];
      s2 += part[((int64_t)gi * 2 + 1) * D + col];
    }
    red[(slice * 2 + 0) * D + col] = s1;
    red[(slice * 2 + 1) * D + col] = s2;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float t1 = 0.0f, t2 = 0.0f;
    for (int s = 0; s < slices; ++s) {
      t1 += red[(s * 2 + 0) * D + c];
      t2 += red[(s * 2 + 1) * D + c];
    }
    tot[c] = t1;
    tot[D + c] = t2;
  }
  __syncthreads();
}

// per-block partial (sum, sumsq) of y: threads = columns x row slices
__global__ __launch_bounds__(256) void bn_stats_kernel(BnArgs a) {
  float* red = feta_lds;  // [slices][2][D]
  const int D = a.D;
  const int slices = 256 / D > 0 ? 256 / D : 1;
  const int col = threadIdx.x % D, slice = threadIdx.x / D;
  const bool active = (int)threadIdx.x < slices * D;
  const int nrb = (a.M + kRowsPerBlock - 1) / kRowsPerBlock;
  float s1 = 0.0f, s2 = 0.0f;
  if (active) {
    for (int rb = blockIdx.x; rb < nrb; rb += a.G) {
      const int r0 = rb * kRowsPerBlock;
      for (int r = r0 + slice; r < min(r0 + kRowsPerBlock, a.M); r += slices) {
        const float v = a.y[(int64_t)r * D + col];
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
  bn_reduce_partials(a.stats, a.G, D, red, tot);
  for (int c = threadIdx.x; c < D; c += 256) {
    const float mean = tot[c] / (float)a.M;
    const float var = fmaxf(tot[D + c] / (float)a.M - mean * mean, 0.0f);
    const float rstd = rsqrtf(var + a.eps);
    const float scale = a.gamma[c] * rstd;
    sc[c] = scale;
    sh[c] = a.beta[c] - mean * scale;
    if (blockIdx.x == 0) {
      a.mean_rstd[c] = mean;
      a.mean_rstd[D + c] = rstd;
      if (a.running_mean != nullptr) {
        const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
        a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * mean;
        a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * unbiased;
      }
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
  float s1 = 0.0f, s2 = 0.0f;
  if (active) {
    const float mean = a.mean_rstd_in[col], rstd = a.mean_rstd_in[D + col];
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
  bn_reduce_partials(a.partial, a.G, D, red, tot);
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

// tiles per workgroup such that the staged weight slice (rows x (16 tg + 4) floats... ) fits
int tiles_fwd(int KI) {  // LDS: 16 tg (KI + 4) floats
  int tg = 4;
  while (tg > 1 && 16 * tg * (KI + 4) * 4 > 56 * 1024) --tg;
  return tg;
}
int tiles_dx(int NO) {   // LDS: NO (16 tg + 4) floats
  int tg = 4;
  while (tg > 1 && NO * (16 * tg + 4) * 4 > 56 * 1024) --tg;
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
void launch_bwd_ki(const RowLinArgs& a, int grid, size_t lds, hipStream_t stream) {
#define CALL(NOV)                                                               \
  {                                                                             \
    auto kern = rowlin_bwd_kernel<KI, NOV>;                                     \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kRowThreads), lds, stream, a);    \
  }
  FETA_DIM_SWITCH(a.NO, CALL)
#undef CALL
}

}  // namespace feta

using namespace feta;

extern "C" int feta_rowlin_blocks(int M) { return row_blocks(M); }
extern "C" int feta_rowlin_chunks(int M) { return row_chunks(M); }

extern "C" int feta_rowlin_fwd(const float* x, const float* w, const float* bias,
                               const float* rowscale, const float* residual, float* y, float* stats,
                               int relu, int M, int KI, int NO, feta_stream_t stream) {
  FETA_REQUIRE(x && w && y && M > 0, "rowlin_fwd: null pointer / empty");
  FETA_REQUIRE(dim_ok(KI) && (NO % 16) == 0 && NO > 0 && NO <= 1024,
               "rowlin_fwd: unsupported dims KI=%d NO=%d", KI, NO);
  FETA_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && (!residual || aligned16(residual)),
               "rowlin_fwd: pointers must be 16-byte aligned");
  RowLinArgs a{};
  a.x = x; a.w = w; a.bias = bias; a.rowscale = rowscale; a.residual = residual; a.y = y;
  a.stats = stats; a.relu = relu; a.M = M; a.KI = KI; a.NO = NO; a.G = row_blocks(M);
  a.TG = tiles_fwd(KI);
  const int n_og = (NO / 16 + a.TG - 1) / a.TG;
  const size_t lds = sizeof(float) * (16 * a.TG * (KI + 4) + (stats ? kRowWaves * 2 * 16 * a.TG : 0));
#define CALL(KV)                                                                                \
  {                                                                                             \
    auto kern = rowlin_fwd_kernel<KV>;                                                          \
    hipLaunchKernelGGL(kern, dim3(a.G * n_og), dim3(kRowThreads), lds, (hipStream_t)stream, a); \
  }
  FETA_DIM_SWITCH(KI, CALL)
#undef CALL
  return check_launch("feta_rowlin_fwd");
}

extern "C" int feta_rowlin_bwd(const float* x, const float* w, const float* dy,
                               const float* rowscale, const float* ysaved, float* dx, float* partial,
                               float* dwdb, int M, int KI, int NO, feta_stream_t stream) {
  FETA_REQUIRE(x && w && dy && dx && partial && dwdb && M > 0, "rowlin_bwd: null pointer / empty");
  FETA_REQUIRE(dim_ok(KI) && dim_ok(NO), "rowlin_bwd: unsupported dims KI=%d NO=%d", KI, NO);
  FETA_REQUIRE(aligned16(x) && aligned16(w) && aligned16(dy) && aligned16(dx) &&
                   (!ysaved || aligned16(ysaved)),
               "rowlin_bwd: pointers must be 16-byte aligned");
  RowLinArgs a{};
  a.x = x; a.w = w; a.dy = dy; a.rowscale = rowscale; a.ysaved = ysaved; a.dx = dx;
  a.partial = partial; a.M = M; a.KI = KI; a.NO = NO;
  a.RC = row_chunks(M);
  a.G = row_blocks(M);
  a.TG = tiles_dx(NO);
  const int n_kg = (KI / 16 + a.TG - 1) / a.TG;
  a.dx_blocks = a.G * n_kg;
  const int dw_blocks = (a.RC * (NO / 16) + kRowWaves - 1) / kRowWaves;
  const int grid = a.dx_blocks + dw_blocks;
  const size_t lds = sizeof(float) * NO * (16 * a.TG + 4);
#define CALL(KV) launch_bwd_ki<KV>(a, grid, lds, (hipStream_t)stream);
  FETA_DIM_SWITCH(KI, CALL)
#undef CALL
  int rc = check_launch("feta_rowlin_bwd");
  if (rc != FETA_OK) return rc;
  return feta_colsum(partial, dwdb, a.RC, NO * KI + NO, stream);
}

extern "C" int feta_bn_stats(const float* y, float* stats, int M, int D, feta_stream_t stream) {
  FETA_REQUIRE(y && stats && M > 0 && D > 0 && D <= 256, "bn_stats: need 0 < D <= 256");
  BnArgs a{};
  a.y = y; a.stats_out = stats; a.M = M; a.D = D; a.G = row_blocks(M);
  const int slices = 256 / D > 0 ? 256 / D : 1;
  auto kern = bn_stats_kernel;
  hipLaunchKernelGGL(kern, dim3(a.G), dim3(256), sizeof(float) * slices * 2 * D, (hipStream_t)stream, a);
  return check_launch("feta_bn_stats");
}

extern "C" int feta_bn_apply_fwd(const float* y, const float* stats, const float* gamma,
                                 const float* beta, float* out, float* mean_rstd, float* running_mean,
                                 float* running_var, float momentum, float eps, int M, int D,
                                 feta_stream_t stream) {
  FETA_REQUIRE(y && stats && gamma && beta && out && mean_rstd, "bn_apply_fwd: null pointer");
  FETA_REQUIRE(M > 0 && D > 0 && D <= 256 && (D % 4) == 0, "bn_apply_fwd: need D %% 4 == 0, D <= 256");
  FETA_REQUIRE(aligned16(y) && aligned16(out), "bn_apply_fwd: pointers must be 16-byte aligned");
  BnArgs a{};
  a.y = y; a.stats = stats; a.gamma = gamma; a.beta = beta; a.out = out; a.mean_rstd = mean_rstd;
  a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum; a.eps = eps;
  a.M = M; a.D = D; a.G = row_blocks(M);
  const int64_t n4 = (int64_t)M * D / 4;
  int grid = (int)((n4 + 255) / 256);
  grid = grid > 1024 ? 1024 : grid;
  const int slices = 256 / D > 0 ? 256 / D : 1;
  auto kern = bn_apply_fwd_kernel;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), sizeof(float) * (4 + 2 * slices) * D,
                     (hipStream_t)stream, a);
  return check_launch("feta_bn_apply_fwd");
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
  hipLaunchKernelGGL(k2, dim3(grid), dim3(256), sizeof(float) * (5 + 2 * slices) * D,
                     (hipStream_t)stream, a);
  return check_launch("feta_bn_bwd");
}
