// LayerNorm of a 64-wide row "on load" (ABI 9): LayerNorm is row-local, so the kernels of the layer stack apply it when
// they stage a row - forward (x_ln_gamma: the consumer normalises the pre-norm rows of its producer) and backward
// (ln*_gamma: the LayerNorm backward of a gradient row, from the pre-norm row it belongs to) - and no normalised
// activation and no LayerNorm launch exists between the fused kernels (norm1 / norm2 of DiffTransformerEncoderLayer with
// batch_norm=False, the default of the reference's TU / molhiv / SBM scripts: experiments/run_transformer_gengcn_cv.py:56).
// Arithmetic of torch.nn.functional.layer_norm: mean and BIASED variance of the row, two passes (the row is in
// registers), rstd = 1 / sqrt(var + eps).
#pragma once
#include "feta_lp.h"

namespace feta {

constexpr int kLnD = 64;

// sum over the RV = 64 / VEC consecutive lanes that hold one row, VEC elements each (fp32 rows: 16 lanes = one DPP row;
// bf16 rows: 8 lanes = half a DPP row); every lane of the group gets the total
template <int VEC>
__device__ __forceinline__ float ln_lanes_sum(float v);
template <>
__device__ __forceinline__ float ln_lanes_sum<4>(float v) { return row16_sum(v); }
template <>
__device__ __forceinline__ float ln_lanes_sum<8>(float v) { return row8_sum(v); }

// mean and rstd of the row whose VEC-element chunk this lane holds
template <int VEC>
__device__ __forceinline__ void ln_row_stats(const float (&f)[VEC], float eps, float& mean, float& rstd) {
  float s = 0.0f;
#pragma unroll
  for (int e = 0; e < VEC; ++e) s += f[e];
  mean = ln_lanes_sum<VEC>(s) * (1.0f / kLnD);
  float q = 0.0f;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float d = f[e] - mean;
    q += d * d;
  }
  rstd = 1.0f / sqrtf(ln_lanes_sum<VEC>(q) * (1.0f / kLnD) + eps);
}

// f <- LayerNorm(row) * gamma + beta for this lane's chunk (gamma, beta: the chunk's VEC columns)
template <int VEC>
__device__ __forceinline__ void ln_apply(float (&f)[VEC], const float* gamma, const float* beta, float eps) {
  float mean, rstd;
  ln_row_stats<VEC>(f, eps, mean, rstd);
#pragma unroll
  for (int e = 0; e < VEC; ++e) f[e] = (f[e] - mean) * rstd * gamma[e] + beta[e];
}

// LayerNorm backward of a gradient row: dv = gradient w.r.t. LN(y) * gamma + beta (this lane's chunk), yv = the pre-norm
// row chunk.  On return dv = gradient w.r.t. y; dgam / dbet accumulate dv_in * xhat / dv_in (the lane's columns).
// keep = false: the row does not exist (padding of a tile) - contributes nothing and comes back as zero.
template <int VEC>
__device__ __forceinline__ void ln_backward(float (&dv)[VEC], const float (&yv)[VEC], const float* gamma, float eps,
                                            bool keep, float (&dgam)[VEC], float (&dbet)[VEC]) {
  float mean, rstd;
  ln_row_stats<VEC>(yv, eps, mean, rstd);
  float xh[VEC], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float d = keep ? dv[e] : 0.0f;
    xh[e] = (yv[e] - mean) * rstd;
    dgam[e] += d * xh[e];
    dbet[e] += d;
    dv[e] = d * gamma[e];
    s1 += dv[e];
    s2 += dv[e] * xh[e];
  }
  s1 = ln_lanes_sum<VEC>(s1) * (1.0f / kLnD);
  s2 = ln_lanes_sum<VEC>(s2) * (1.0f / kLnD);
#pragma unroll
  for (int e = 0; e < VEC; ++e) dv[e] = keep ? rstd * (dv[e] - s1 - xh[e] * s2) : 0.0f;
}

}  // namespace feta
