// Storage policy of the fused layer-stack kernels (block.hip, block_bwd.hip, ffn.hip, ffn_bwd.hip): the kernels are
// written ONCE against T in {float, bf16_t}.
//
//   T = float   the reference's arithmetic: token tensors and LDS tiles fp32, every contraction a chain of k-ordered
//               v_mfma_f32_16x16x4_f32 (exact fp32) - what these kernels were before they became templates;
//   T = bf16_t  BASELINE configs 3 / 5 (the reference has no reduced-precision mode, experiments/
//               run_transformer_gengcn.py:115-164): token tensors (x, qkv, out, y, h, their gradients) and pe live in
//               HBM as bf16, the LDS tiles hold bf16 (half the LDS bytes, 8-byte operand reads), each group of four
//               k-steps is ONE v_mfma_f32_16x16x16_bf16 (8 cycles of the matrix pipe against 4 x 32), and an operand
//               is rounded once - when it is staged or when an accumulator is handed on - however many products it
//               enters.  fp32: master weights in HBM (rounded when they are staged), every accumulator, softmax and
//               BatchNorm statistics, partial sums of weight / bias / affine gradients, the attention matrix handed to
//               the coefficient generator.
//
// With the lane layout of feta_tiles.h (A[row lq][k = g], four steps s covering k = 4g + s) the four values a lane
// feeds to four f32 MFMAs are the four k-consecutive values it feeds to one bf16 MFMA: no operand changes place
// between the two instantiations (feta_bf16.h makes the same observation for the general kernels).
#pragma once
#include "feta_tiles.h"

namespace feta {

template <class T>
struct Lp;

template <>
struct Lp<float> {
  static constexpr int VEC = 4;   // elements of a 16-byte vector
  static constexpr int PAD = 4;   // LDS row padding in elements: pitch = 4 * odd dwords, 16-byte operand reads of the
                                  // two 4-row groups of a half-wave hit disjoint banks
  struct Op {
    float v[4];
  };
  struct Vec {
    float4 raw;
  };
  static __device__ __forceinline__ Op zero() { return Op{{0.0f, 0.0f, 0.0f, 0.0f}}; }
  static __device__ __forceinline__ Op mk(float a, float b, float c, float d) { return Op{{a, b, c, d}}; }
  static __device__ __forceinline__ Op mk(const f32x4& a) { return Op{{a[0], a[1], a[2], a[3]}}; }
  static __device__ __forceinline__ Op ld(const float* p) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    return Op{{x.x, x.y, x.z, x.w}};
  }
  // an operand parked in LDS in its register layout (16 bytes per lane)
  static __device__ __forceinline__ Op ldo(const Op* p) { return ld(reinterpret_cast<const float*>(p)); }
  static __device__ __forceinline__ void sto(Op* p, const Op& o) {
    *reinterpret_cast<float4*>(p) = make_float4(o.v[0], o.v[1], o.v[2], o.v[3]);
  }
  static __device__ __forceinline__ Op ld_scaled(const float* p, float s) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    return Op{{x.x * s, x.y * s, x.z * s, x.w * s}};
  }
  // p[0], p[stride], p[2 stride], p[3 stride]: an operand whose k runs down a column of a row-major tile
  static __device__ __forceinline__ Op gather(const float* p, int stride) {
    return Op{{p[0], p[stride], p[2 * stride], p[3 * stride]}};
  }
  static __device__ __forceinline__ float get(const Op& o, int i) { return o.v[i]; }
  static __device__ __forceinline__ Op sel(bool c, const Op& a, const Op& b) {
    return Op{{c ? a.v[0] : b.v[0], c ? a.v[1] : b.v[1], c ? a.v[2] : b.v[2], c ? a.v[3] : b.v[3]}};
  }
  static __device__ __forceinline__ f32x4 mma(const Op& a, const Op& b, f32x4 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma16(a.v[s], b.v[s], acc);
    return acc;
  }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ float to_f(float v) { return v; }     // one stored element as it arrived -> fp32
  static __device__ __forceinline__ float one() { return 1.0f; }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  }
  static __device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
  static __device__ __forceinline__ Vec ldv(const float* p) { return Vec{*reinterpret_cast<const float4*>(p)}; }
  static __device__ __forceinline__ void stv(float* p, const Vec& v) { *reinterpret_cast<float4*>(p) = v.raw; }
  static __device__ __forceinline__ void unpack(const Vec& v, float (&f)[VEC]) {
    f[0] = v.raw.x; f[1] = v.raw.y; f[2] = v.raw.z; f[3] = v.raw.w;
  }
  static __device__ __forceinline__ Vec pack(const float (&f)[VEC]) { return Vec{make_float4(f[0], f[1], f[2], f[3])}; }
};

template <>
struct Lp<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr int PAD = 8;   // pitch = 4 * odd dwords: the 8-byte operand reads of a half-wave cover all 64 banks
  typedef bf16x4_pk Op;
  struct __attribute__((aligned(16))) Vec {
    bf16x4_pk lo, hi;
  };
  static __device__ __forceinline__ Op zero() { return pack_bf16x4(0.0f, 0.0f, 0.0f, 0.0f); }
  static __device__ __forceinline__ Op mk(float a, float b, float c, float d) { return pack_bf16x4(a, b, c, d); }
  static __device__ __forceinline__ Op mk(const f32x4& a) { return pack_bf16x4(a[0], a[1], a[2], a[3]); }
  static __device__ __forceinline__ Op ld(const bf16_t* p) { return *reinterpret_cast<const bf16x4_pk*>(p); }
  static __device__ __forceinline__ Op ldo(const Op* p) { return *p; }
  static __device__ __forceinline__ void sto(Op* p, const Op& o) { *p = o; }
  static __device__ __forceinline__ Op ld_scaled(const bf16_t* p, float s) {
    const Op x = ld(p);
    return pack_bf16x4(bf16x4_get(x, 0) * s, bf16x4_get(x, 1) * s, bf16x4_get(x, 2) * s, bf16x4_get(x, 3) * s);
  }
  static __device__ __forceinline__ Op gather(const bf16_t* p, int stride) {
    return pack_bf16x4_raw(p[0], p[stride], p[2 * stride], p[3 * stride]);
  }
  static __device__ __forceinline__ float get(const Op& o, int i) { return bf16x4_get(o, i); }
  static __device__ __forceinline__ Op sel(bool c, const Op& a, const Op& b) {
    Op r;
    r.v[0] = c ? a.v[0] : b.v[0]; r.v[1] = c ? a.v[1] : b.v[1]; r.v[2] = c ? a.v[2] : b.v[2]; r.v[3] = c ? a.v[3] : b.v[3];
    return r;
  }
  static __device__ __forceinline__ f32x4 mma(const Op& a, const Op& b, f32x4 acc) { return mfma16_bf16_pk(a, b, acc); }
  static __device__ __forceinline__ float ld1(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
  static __device__ __forceinline__ bf16_t one() { return f2bf(1.0f); }
  static __device__ __forceinline__ void st1(bf16_t* p, float v) { *p = f2bf(v); }
  static __device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
    const Op x = ld(p);
    v[0] = bf16x4_get(x, 0); v[1] = bf16x4_get(x, 1); v[2] = bf16x4_get(x, 2); v[3] = bf16x4_get(x, 3);
  }
  static __device__ __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<bf16x4_pk*>(p) = pack_bf16x4(a, b, c, d);
  }
  static __device__ __forceinline__ Vec ldv(const bf16_t* p) { return *reinterpret_cast<const Vec*>(p); }
  static __device__ __forceinline__ void stv(bf16_t* p, const Vec& v) { *reinterpret_cast<Vec*>(p) = v; }
  static __device__ __forceinline__ void unpack(const Vec& v, float (&f)[VEC]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[i] = bf16x4_get(v.lo, i);
      f[4 + i] = bf16x4_get(v.hi, i);
    }
  }
  static __device__ __forceinline__ Vec pack(const float (&f)[VEC]) {
    Vec v;
    v.lo = pack_bf16x4(f[0], f[1], f[2], f[3]);
    v.hi = pack_bf16x4(f[4], f[5], f[6], f[7]);
    return v;
  }
};

// Row operand for a contraction over DH features of a row of T (LDS tile or global): chunk j covers features
// 16 j + 4 g .. + 3 (the bijection of feta_tiles.h)
template <class T, int DH>
struct RowOp {
  static constexpr int NJ = (DH + 15) / 16;
  typename Lp<T>::Op o[NJ];
};

template <class T, int DH>
__device__ __forceinline__ void load_row_op(RowOp<T, DH>& t, const T* row, int g) {
#pragma unroll
  for (int j = 0; j < RowOp<T, DH>::NJ; ++j) t.o[j] = Lp<T>::ld(row + 16 * j + 4 * g);
}
template <class T, int DH>
__device__ __forceinline__ void load_row_op_scaled(RowOp<T, DH>& t, const T* row, int g, float scale) {
#pragma unroll
  for (int j = 0; j < RowOp<T, DH>::NJ; ++j) t.o[j] = Lp<T>::ld_scaled(row + 16 * j + 4 * g, scale);
}
template <class T, int DH>
__device__ __forceinline__ f32x4 dot_row_ops(const RowOp<T, DH>& a, const RowOp<T, DH>& b, f32x4 acc) {
#pragma unroll
  for (int j = 0; j < RowOp<T, DH>::NJ; ++j) acc = Lp<T>::mma(a.o[j], b.o[j], acc);
  return acc;
}

// the single dynamic-LDS array seen as bytes (tiles of T and fp32 scratch share it)
__device__ __forceinline__ char* lds_bytes() { return reinterpret_cast<char*>(feta_lds); }

}  // namespace feta
