// 16x16 tile operands for v_mfma_f32_16x16x4_f32 on "token tensors"
// (element (b,i,h,c) at p[b*sb + i*sn + h*dh + c]).
//
// Lane naming used by every kernel: lq = lane & 15, g = lane >> 4.
//   accumulator register r of a tile  <->  (row 4g + r, column lq)
//   A operand of one MFMA              <->  A[row lq][k = g]
//   B operand of one MFMA              <->  B[k = g][column lq]
// An accumulator register r can therefore be fed back as the A operand (rows
// become the contraction index k = g <-> row 4g + r, lq stays the row of the new
// product) or as the B operand (k = g <-> row 4g + r, lq the column): products
// that contract over the ROW index of a previous result need no data movement.
//
// Contraction over the feature dim c of a row-major row uses the bijection
// c = 16 j + 4 g + s (chunk j, MFMA step s) so that each lane reads its four
// values as ONE aligned 16-byte load.
#pragma once
#include <feta_device.h>

namespace feta {

template <int DH>
struct Feat {
  static constexpr int NJ = (DH + 15) / 16;  // 16-wide feature chunks
  static constexpr int CT = (DH + 15) / 16;  // 16-wide output column tiles
  float f[NJ][4];
};

// Row operand for a contraction over features: lane reads row[16j + 4g .. +3].
template <int DH>
__device__ __forceinline__ void load_row(Feat<DH>& t, const float* row, int g, float scale = 1.0f) {
#pragma unroll
  for (int j = 0; j < Feat<DH>::NJ; ++j) {
    const int c = 16 * j + 4 * g;
    if (row != nullptr && c < DH) {
      const float4 x = *reinterpret_cast<const float4*>(row + c);
      t.f[j][0] = x.x * scale;
      t.f[j][1] = x.y * scale;
      t.f[j][2] = x.z * scale;
      t.f[j][3] = x.w * scale;
    } else {
      t.f[j][0] = t.f[j][1] = t.f[j][2] = t.f[j][3] = 0.0f;
    }
  }
}

// Same operand, but the load is UNCONDITIONAL (row must point at readable memory - clamp the
// row index) and `ok` only selects zero afterwards: no branch around the load, so the compiler
// can issue a whole batch of such loads back to back and expose ONE memory latency.
template <int DH>
__device__ __forceinline__ void load_row_sel(Feat<DH>& t, const float* row, bool ok, int g,
                                             float scale = 1.0f) {
#pragma unroll
  for (int j = 0; j < Feat<DH>::NJ; ++j) {
    const int c = 16 * j + 4 * g;
    const float4 x = *reinterpret_cast<const float4*>(row + (c < DH ? c : 0));
    const bool sel = ok && c < DH;
    t.f[j][0] = sel ? x.x * scale : 0.0f;
    t.f[j][1] = sel ? x.y * scale : 0.0f;
    t.f[j][2] = sel ? x.z * scale : 0.0f;
    t.f[j][3] = sel ? x.w * scale : 0.0f;
  }
}

// acc += A_rows . B_rows^T over the feature dim: result row index = a's row (lq of
// the lane that loaded it), column index = b's row.
template <int DH>
__device__ __forceinline__ f32x4 dot_rows(const Feat<DH>& a, const Feat<DH>& b, f32x4 acc) {
#pragma unroll
  for (int j = 0; j < Feat<DH>::NJ; ++j) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma16(a.f[j][s], b.f[j][s], acc);
  }
  return acc;
}

__device__ __forceinline__ f32x4 zero4() {
  f32x4 z;
  z[0] = 0.0f;
  z[1] = 0.0f;
  z[2] = 0.0f;
  z[3] = 0.0f;
  return z;
}

__device__ __forceinline__ const float* tok_row(const float* p, int64_t sb, int64_t sn, int b,
                                                int i, int h, int dh) {
  return p + (int64_t)b * sb + (int64_t)i * sn + (int64_t)h * dh;
}
__device__ __forceinline__ float* tok_row(float* p, int64_t sb, int64_t sn, int b, int i, int h,
                                          int dh) {
  return p + (int64_t)b * sb + (int64_t)i * sn + (int64_t)h * dh;
}

}  // namespace feta
