// bf16 STORAGE for the attention core and the eigenbasis filter (BASELINE configs 3 and 5): token tensors,
// pe, U, attn and the per-block filter weights live in HBM as bf16, the contractions run on the bf16 matrix
// pipe (v_mfma_f32_16x16x16_bf16, fp32 accumulate), and everything with a cancellation in it - softmax
// statistics, t_k(lambda), accumulators, reductions - stays fp32.
//
// The kernels (bf16.hip) are written once against a storage type T in {float, bf16_t}: T = float is the
// arithmetic of attn.hip / filter.hip's general kernels (groups of four k-ordered v_mfma_f32_16x16x4_f32),
// T = bf16_t rounds the operands of every contraction to bf16 and issues ONE 16x16x16 MFMA per group:
// with the lane layout of feta_tiles.h (A[row lq][k = g], four steps s = 0..3 covering k = 4g + s) the
// four values a lane feeds to four f32 MFMAs are exactly the four k-consecutive bf16 values it feeds to
// one bf16 MFMA, so no operand changes place.
#pragma once
#include "feta_tiles.h"

namespace feta {

template <class T>
struct Num;

template <>
struct Num<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  // four consecutive elements, 16-byte aligned
  static __device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  }
  static __device__ __forceinline__ f32x4 mma4(const float (&a)[4], const float (&b)[4], f32x4 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma16(a[s], b[s], acc);
    return acc;
  }
};

template <>
struct Num<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  // four consecutive elements, 8-byte aligned
  static __device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
    const bf16x4_raw x = *reinterpret_cast<const bf16x4_raw*>(p);
    v[0] = bf2f(x.v[0]); v[1] = bf2f(x.v[1]); v[2] = bf2f(x.v[2]); v[3] = bf2f(x.v[3]);
  }
  static __device__ __forceinline__ f32x4 mma4(const float (&a)[4], const float (&b)[4], f32x4 acc) {
    return mfma16_bf16(a, b, acc);
  }
};

// Row operand for a contraction over features (load_row_sel of feta_tiles.h for a storage type)
template <class T, int DH>
__device__ __forceinline__ void load_row_t(Feat<DH>& t, const T* row, bool ok, int g, float scale = 1.0f) {
#pragma unroll
  for (int j = 0; j < Feat<DH>::NJ; ++j) {
    const int c = 16 * j + 4 * g;
    float x[4];
    Num<T>::ld4(row + (c < DH ? c : 0), x);
    const bool sel = ok && c < DH;
#pragma unroll
    for (int s = 0; s < 4; ++s) t.f[j][s] = sel ? x[s] * scale : 0.0f;
  }
}

template <class T, int DH>
__device__ __forceinline__ f32x4 dot_rows_t(const Feat<DH>& a, const Feat<DH>& b, f32x4 acc) {
#pragma unroll
  for (int j = 0; j < Feat<DH>::NJ; ++j) acc = Num<T>::mma4(a.f[j], b.f[j], acc);
  return acc;
}

template <class T>
__device__ __forceinline__ f32x4 mma_acc(const f32x4& a, const f32x4& b, f32x4 acc) {
  const float av[4] = {a[0], a[1], a[2], a[3]}, bv[4] = {b[0], b[1], b[2], b[3]};
  return Num<T>::mma4(av, bv, acc);
}

template <class T>
__device__ __forceinline__ const T* tok_row_t(const T* p, int64_t sb, int64_t sn, int b, int i, int h, int dh) {
  return p + (int64_t)b * sb + (int64_t)i * sn + (int64_t)h * dh;
}
template <class T>
__device__ __forceinline__ T* tok_row_t(T* p, int64_t sb, int64_t sn, int b, int i, int h, int dh) {
  return p + (int64_t)b * sb + (int64_t)i * sn + (int64_t)h * dh;
}

}  // namespace feta
