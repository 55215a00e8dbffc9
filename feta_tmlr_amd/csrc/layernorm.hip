// LayerNorm over the feature dimension of the [M, D] row view: norm1 / norm2 of
// DiffTransformerEncoderLayer when batch_norm=False (contract transformer/models.py:505-506; the
// reference's --batch-norm flag is off by default for the TU, molhiv and SBM scripts,
// experiments/run_transformer_gengcn_cv.py:56).  A row is owned by 16 lanes (one DPP row, 16 bytes per
// lane and 64 columns), so mean, variance and the two backward sums are row16_sum on the VALU; a
// workgroup of 256 threads walks row blocks of 16 and keeps the dgamma / dbeta column partials in
// registers until its last block.
#include "feta_abi_common.h"
#include "feta_lp.h"

namespace feta {

constexpr int kLnThreads = 256, kLnRows = kLnThreads / 16, kLnMaxBlocks = 256, kLnMaxV = 4;

inline int ln_blocks(int M) {
  const int nb = (M + kLnRows - 1) / kLnRows;
  return nb < kLnMaxBlocks ? nb : kLnMaxBlocks;
}

struct LnArgs {
  const float* y;       // [M, D] input rows
  const float* gamma;   // [D]
  const float* beta;    // [D]
  float* out;           // [M, D]
  float* stats;         // [M, 2] mean, rstd
  const float* dout;    // [M, D]
  float* dy;            // [M, D]
  float* partial;       // [gridDim.x, 2, D] dgamma | dbeta partial sums (row pitch partial_ld)
  float eps;
  int M, D, partial_ld;
  int y_bf16, out_bf16, dout_bf16, dy_bf16;   // storage type per tensor (feta_layernorm_*_ex): 1 = bf16
};

// four consecutive elements of a row of fp32 or bf16 storage (the pointer is typed float* in LnArgs either way)
__device__ __forceinline__ float4 ln_ld4(const float* base, int64_t idx, bool bf16) {
  float v[4];
  if (bf16) Lp<bf16_t>::ld4(reinterpret_cast<const bf16_t*>(base) + idx, v);
  else Lp<float>::ld4(base + idx, v);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void ln_st4(float* base, int64_t idx, bool bf16, const float4& o) {
  if (bf16) Lp<bf16_t>::st4(reinterpret_cast<bf16_t*>(base) + idx, o.x, o.y, o.z, o.w);
  else Lp<float>::st4(base + idx, o.x, o.y, o.z, o.w);
}

template <int NV>
__global__ __launch_bounds__(kLnThreads) void ln_fwd_kernel(LnArgs a) {
  const int l = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const float inv_d = 1.0f / (float)a.D;
  float4 gm[NV], bt[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = min(64 * v + 4 * l, a.D - 4);
    gm[v] = *reinterpret_cast<const float4*>(a.gamma + c);
    bt[v] = *reinterpret_cast<const float4*>(a.beta + c);
  }
  const int nblk = (a.M + kLnRows - 1) / kLnRows;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int row = blk * kLnRows + rg;
    const int rowc = min(row, a.M - 1);
    float4 x[NV];
    float s = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = 64 * v + 4 * l;
      const bool ok = c < a.D;
      x[v] = ln_ld4(a.y, (int64_t)rowc * a.D + (ok ? c : 0), a.y_bf16 != 0);
      if (!ok) x[v].x = x[v].y = x[v].z = x[v].w = 0.0f;
      s += (x[v].x + x[v].y) + (x[v].z + x[v].w);
    }
    const float mean = row16_sum(s) * inv_d;
    float q = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (64 * v + 4 * l < a.D) {
        const float dx = x[v].x - mean, dy = x[v].y - mean, dz = x[v].z - mean, dw = x[v].w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    const float rstd = 1.0f / sqrtf(row16_sum(q) * inv_d + a.eps);
    if (row < a.M) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = 64 * v + 4 * l;
        if (c < a.D) {
          float4 o;
          o.x = (x[v].x - mean) * rstd * gm[v].x + bt[v].x;
          o.y = (x[v].y - mean) * rstd * gm[v].y + bt[v].y;
          o.z = (x[v].z - mean) * rstd * gm[v].z + bt[v].z;
          o.w = (x[v].w - mean) * rstd * gm[v].w + bt[v].w;
          ln_st4(a.out, (int64_t)row * a.D + c, a.out_bf16 != 0, o);
        }
      }
      if (l == 0) {
        a.stats[2 * (int64_t)row] = mean;
        a.stats[2 * (int64_t)row + 1] = rstd;
      }
    }
  }
}

// dy = rstd (g - mean_c(g) - xhat mean_c(g xhat)),  g = dout gamma,  xhat = (y - mean) rstd;
// dgamma = sum_rows dout xhat, dbeta = sum_rows dout (per-workgroup partials, reduced by feta_colsum)
template <int NV>
__global__ __launch_bounds__(kLnThreads) void ln_bwd_kernel(LnArgs a) {
  const int l = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const float inv_d = 1.0f / (float)a.D;
  float4 gm[NV], dg[NV], db[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gm[v] = *reinterpret_cast<const float4*>(a.gamma + min(64 * v + 4 * l, a.D - 4));
    dg[v] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    db[v] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  const int nblk = (a.M + kLnRows - 1) / kLnRows;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int row = blk * kLnRows + rg;
    const int rowc = min(row, a.M - 1);
    const bool rok = row < a.M;
    float4 yr[NV], dr[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = 64 * v + 4 * l;
      yr[v] = ln_ld4(a.y, (int64_t)rowc * a.D + (c < a.D ? c : 0), a.y_bf16 != 0);
      dr[v] = ln_ld4(a.dout, (int64_t)rowc * a.D + (c < a.D ? c : 0), a.dout_bf16 != 0);
    }
    float mean, rstd;
    if (a.stats != nullptr) {
      mean = a.stats[2 * (int64_t)rowc];
      rstd = a.stats[2 * (int64_t)rowc + 1];
    } else {
      // no saved statistics (the forward applied this LayerNorm on load, csrc/feta_ln.h): the row is here - two more sums
      float s0 = 0.0f, q0 = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (64 * v + 4 * l < a.D) s0 += (yr[v].x + yr[v].y) + (yr[v].z + yr[v].w);
      mean = row16_sum(s0) * inv_d;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (64 * v + 4 * l < a.D) {
          const float dx = yr[v].x - mean, dy_ = yr[v].y - mean, dz = yr[v].z - mean, dw = yr[v].w - mean;
          q0 += (dx * dx + dy_ * dy_) + (dz * dz + dw * dw);
        }
      rstd = 1.0f / sqrtf(row16_sum(q0) * inv_d + a.eps);
    }
    float4 xh[NV], g[NV];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = 64 * v + 4 * l;
      const bool ok = c < a.D && rok;
      const float4 yv = yr[v];
      const float4 dv = dr[v];
      const float m = ok ? 1.0f : 0.0f;
      xh[v].x = (yv.x - mean) * rstd * m;  xh[v].y = (yv.y - mean) * rstd * m;
      xh[v].z = (yv.z - mean) * rstd * m;  xh[v].w = (yv.w - mean) * rstd * m;
      const float4 d = make_float4(dv.x * m, dv.y * m, dv.z * m, dv.w * m);
      g[v].x = d.x * gm[v].x;  g[v].y = d.y * gm[v].y;  g[v].z = d.z * gm[v].z;  g[v].w = d.w * gm[v].w;
      s1 += (g[v].x + g[v].y) + (g[v].z + g[v].w);
      s2 += (g[v].x * xh[v].x + g[v].y * xh[v].y) + (g[v].z * xh[v].z + g[v].w * xh[v].w);
      dg[v].x += d.x * xh[v].x;  dg[v].y += d.y * xh[v].y;  dg[v].z += d.z * xh[v].z;  dg[v].w += d.w * xh[v].w;
      db[v].x += d.x;  db[v].y += d.y;  db[v].z += d.z;  db[v].w += d.w;
    }
    s1 = row16_sum(s1) * inv_d;
    s2 = row16_sum(s2) * inv_d;
    if (rok) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = 64 * v + 4 * l;
        if (c < a.D) {
          float4 o;
          o.x = rstd * (g[v].x - s1 - xh[v].x * s2);
          o.y = rstd * (g[v].y - s1 - xh[v].y * s2);
          o.z = rstd * (g[v].z - s1 - xh[v].z * s2);
          o.w = rstd * (g[v].w - s1 - xh[v].w * s2);
          ln_st4(a.dy, (int64_t)row * a.D + c, a.dy_bf16 != 0, o);
        }
      }
    }
  }
  // column partials of the workgroup: [16 row groups][2][D] in LDS, summed in row-group order
  float* red = feta_lds;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = 64 * v + 4 * l;
    if (c < a.D) {
      *reinterpret_cast<float4*>(red + (rg * 2 + 0) * a.D + c) = dg[v];
      *reinterpret_cast<float4*>(red + (rg * 2 + 1) * a.D + c) = db[v];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * a.D; i += kLnThreads) {
    float s = 0.0f;
    for (int r = 0; r < kLnRows; ++r) s += red[r * 2 * a.D + i];
    a.partial[(int64_t)blockIdx.x * a.partial_ld + i] = s;
  }
}

inline bool ln_dim_ok(int D) { return D >= 4 && D <= 64 * kLnMaxV && (D % 4) == 0; }

}  // namespace feta

using namespace feta;

extern "C" int feta_layernorm_blocks(int M) { return M > 0 ? ln_blocks(M) : 0; }

#define FETA_LN_SWITCH(nv, CALL) \
  switch (nv) {                  \
    case 1: CALL(1) break;       \
    case 2: CALL(2) break;       \
    case 3: CALL(3) break;       \
    default: CALL(4) break;      \
  }

extern "C" int feta_layernorm_fwd(const float* y, const float* gamma, const float* beta, float eps, float* out,
                                  float* stats, int M, int D, feta_stream_t stream) {
  return feta_layernorm_fwd_ex(y, gamma, beta, eps, out, stats, M, D, FETA_F32, FETA_F32, stream);
}

extern "C" int feta_layernorm_fwd_ex(const void* y, const float* gamma, const float* beta, float eps, void* out,
                                     float* stats, int M, int D, int y_dtype, int out_dtype, feta_stream_t stream) {
  FETA_REQUIRE(y && gamma && beta && out && stats && M > 0, "layernorm_fwd: bad arguments");
  FETA_REQUIRE(ln_dim_ok(D), "layernorm_fwd: D = %d (multiple of 4, <= 256)", D);
  FETA_REQUIRE((y_dtype == FETA_F32 || y_dtype == FETA_BF16) && (out_dtype == FETA_F32 || out_dtype == FETA_BF16),
               "layernorm_fwd: dtypes %d, %d", y_dtype, out_dtype);
  FETA_REQUIRE(aligned16(y) && aligned16(gamma) && aligned16(beta) && aligned16(out),
               "layernorm_fwd: pointers must be 16-byte aligned");
  LnArgs a{};
  a.y = static_cast<const float*>(y);
  a.gamma = gamma;
  a.beta = beta;
  a.out = static_cast<float*>(out);
  a.stats = stats;
  a.eps = eps;
  a.M = M;
  a.D = D;
  a.y_bf16 = y_dtype == FETA_BF16;
  a.out_bf16 = out_dtype == FETA_BF16;
  const int nblk = (M + kLnRows - 1) / kLnRows;
  const dim3 grid(nblk < 8 * kLnMaxBlocks ? nblk : 8 * kLnMaxBlocks), block(kLnThreads);
#define CALL(NVV) { auto kern = ln_fwd_kernel<NVV>; hipLaunchKernelGGL(kern, grid, block, 0, (hipStream_t)stream, a); }
  FETA_LN_SWITCH((D + 63) / 64, CALL)
#undef CALL
  return check_launch("feta_layernorm_fwd");
}

extern "C" int feta_layernorm_bwd(const float* dout, const float* y, const float* stats, const float* gamma,
                                  float* dy, float* partial, int partial_ld, float* dgamma_dbeta, int M, int D,
                                  feta_stream_t stream) {
  return feta_layernorm_bwd_ex(dout, y, stats, gamma, dy, partial, partial_ld, dgamma_dbeta, M, D, FETA_F32, FETA_F32,
                               FETA_F32, stream);
}

extern "C" int feta_layernorm_bwd_ex(const void* dout, const void* y, const float* stats, const float* gamma,
                                     void* dy, float* partial, int partial_ld, float* dgamma_dbeta, int M, int D,
                                     int dout_dtype, int y_dtype, int dy_dtype, feta_stream_t stream) {
  return feta_layernorm_bwd_eps(dout, y, stats, 1e-5f, gamma, dy, partial, partial_ld, dgamma_dbeta, M, D, dout_dtype, y_dtype,
                                dy_dtype, stream);
}

extern "C" int feta_layernorm_bwd_eps(const void* dout, const void* y, const float* stats, float eps, const float* gamma,
                                      void* dy, float* partial, int partial_ld, float* dgamma_dbeta, int M, int D,
                                      int dout_dtype, int y_dtype, int dy_dtype, feta_stream_t stream) {
  FETA_REQUIRE(dout && y && gamma && dy && partial && M > 0, "layernorm_bwd: bad arguments");
  FETA_REQUIRE(partial_ld == 0 || partial_ld >= 2 * D, "layernorm_bwd: partial_ld = %d < 2 D", partial_ld);
  FETA_REQUIRE(dgamma_dbeta || partial_ld > 0, "layernorm_bwd: dgamma_dbeta may only be NULL with a caller-reduced partial_ld");
  FETA_REQUIRE(ln_dim_ok(D), "layernorm_bwd: D = %d (multiple of 4, <= 256)", D);
  FETA_REQUIRE((dout_dtype | y_dtype | dy_dtype) >= 0 && (dout_dtype | y_dtype | dy_dtype) <= 1,
               "layernorm_bwd: dtypes %d, %d, %d", dout_dtype, y_dtype, dy_dtype);
  FETA_REQUIRE(aligned16(dout) && aligned16(y) && aligned16(gamma) && aligned16(dy),
               "layernorm_bwd: pointers must be 16-byte aligned");
  LnArgs a{};
  a.y = static_cast<const float*>(y);
  a.gamma = gamma;
  a.stats = const_cast<float*>(stats);
  a.eps = eps;
  a.dout = static_cast<const float*>(dout);
  a.dy = static_cast<float*>(dy);
  a.partial = partial;
  a.partial_ld = partial_ld > 0 ? partial_ld : 2 * D;
  a.M = M;
  a.D = D;
  a.dout_bf16 = dout_dtype == FETA_BF16;
  a.y_bf16 = y_dtype == FETA_BF16;
  a.dy_bf16 = dy_dtype == FETA_BF16;
  const int G = ln_blocks(M);
  const size_t lds = sizeof(float) * kLnRows * 2 * D;
#define CALL(NVV) { auto kern = ln_bwd_kernel<NVV>; hipLaunchKernelGGL(kern, dim3(G), dim3(kLnThreads), lds, (hipStream_t)stream, a); }
  FETA_LN_SWITCH((D + 63) / 64, CALL)
#undef CALL
  const int rc = check_launch("feta_layernorm_bwd");
  if (rc != FETA_OK || dgamma_dbeta == nullptr) return rc;   // NULL: the caller reduces all its partials at once
  FETA_REQUIRE(partial_ld == 0, "layernorm_bwd: dgamma_dbeta with partial_ld is not supported");
  return feta_colsum(partial, dgamma_dbeta, G, 2 * D, stream);
}
