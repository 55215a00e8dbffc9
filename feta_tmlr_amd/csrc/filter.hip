// A3: the dynamic spectral filter of ChebConvDynamic, forward and backward, in two
// algebraically equal forms, one wave per (graph b, head h):
//   cheb : direct Chebyshev recursion on the dense scaled Laplacian Lhat_b
//          (transformer/ChebNetDynamic.py:157-187), evaluated by Clenshaw's
//          recurrence on Z_k = X W_k so that every product contracts over the node
//          index held on accumulator rows (no transposes, no LDS round trips);
//   spec : eigenbasis form U [sum_k diag(t_k(lam)) U^T (X W_k)].
// Both fold in the reference's glue: per-(head,graph) weights coeff[h*B+b] reshaped
// [P,dh,dh] (transformer/models.py:357), gather of real nodes (:347), scatter into a
// zero-initialised [N,B,d] tensor (:200-202) and the un-replicated edge_index quirk
// (:186; heads_share_graph = 0 gives heads >= 1 an empty graph, Lhat = 0).
//
// Tile conventions: feta_tiles.h.  "acc layout" of a [rows x cols] tile means register
// r of lane (g, lq) holds element (row 4g + r, column lq).
#include "feta_abi_common.h"
#include "feta_rowops.h"
#include "feta_tiles.h"

namespace feta {

struct FilterArgs {
  const float* x;
  const float* lhat;  // cheb
  const float* u;     // spec
  const float* lam;   // spec
  const float* coeff;
  const float* bias;
  const int32_t* n_real;
  const float* dy;
  float* y;
  float* dx;
  float* dcoeff;
  float* dbias_part;
  int64_t xsb, xsn, ysb, ysn;
  int B, N, H, P, K;
  int share;
  int total;  // B * H
};

constexpr int kFWaves = 4;

// ---- operand helpers -------------------------------------------------------------

// acc-layout load of a token-tensor tile: rows row0 + 4g + r (valid < nvalid), column 16ct + lq
template <int DH>
__device__ __forceinline__ f32x4 load_acc(const float* p, int64_t sb, int64_t sn, int b, int h,
                                          int row0, int nvalid, int ct, int lq, int g) {
  f32x4 v;
  const int c = 16 * ct + lq;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int node = row0 + 4 * g + r;
    v[r] = (node < nvalid && c < DH) ? tok_row(p, sb, sn, b, node, h, DH)[c] : 0.0f;
  }
  return v;
}

// W_k[c][c'] element for the product X . W_k: B[k <-> c = 16j+4g+s][col c' = 16ct+lq]
template <int DH>
__device__ __forceinline__ float w_b(const float* w, int k, int j, int s, int ct, int lq, int g) {
  const int c = 16 * j + 4 * g + s, cp = 16 * ct + lq;
  return (c < DH && cp < DH) ? w[(k * DH + c) * DH + cp] : 0.0f;
}

// W_k[c = 16ct+lq][c' = 16j+4g .. +3] as one 16-byte load (operand of dY . W_k^T and W_k . G)
template <int DH>
__device__ __forceinline__ float4 w_row4(const float* w, int k, int ct, int j, int lq, int g) {
  const int c = 16 * ct + lq, cp = 16 * j + 4 * g;
  if (c < DH && cp < DH) return *reinterpret_cast<const float4*>(w + (k * DH + c) * DH + cp);
  return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// component-wise select (a ternary on a whole float4 can be lowered to a private-memory select)
__device__ __forceinline__ float4 keep4(bool c, const float4& v) {
  return make_float4(c ? v.x : 0.0f, c ? v.y : 0.0f, c ? v.z : 0.0f, c ? v.w : 0.0f);
}

__device__ __forceinline__ float f4(const float4& v, int s) {
  return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w));
}

// Z_k tile = X_tile . W_k  (acc layout [node][c'])
template <int DH>
__device__ __forceinline__ void xw_tile(const Feat<DH>& xf, const float* w, int k, int lq, int g,
                                        f32x4 (&z)[Feat<DH>::CT]) {
  constexpr int CT = Feat<DH>::CT, NJ = Feat<DH>::NJ;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    z[ct] = zero4();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) z[ct] = mfma16(xf.f[j][s], w_b<DH>(w, k, j, s, ct, lq, g), z[ct]);
  }
}

// G_k tile = dY_tile . W_k^T  (acc layout [node][c])
template <int DH>
__device__ __forceinline__ void dyw_tile(const Feat<DH>& dyf, const float* w, int k, int lq, int g,
                                         f32x4 (&gk)[Feat<DH>::CT]) {
  constexpr int CT = Feat<DH>::CT, NJ = Feat<DH>::NJ;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    gk[ct] = zero4();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float4 wv = w_row4<DH>(w, k, ct, j, lq, g);
#pragma unroll
      for (int s = 0; s < 4; ++s) gk[ct] = mfma16(dyf.f[j][s], f4(wv, s), gk[ct]);
    }
  }
}

template <int DH>
__device__ __forceinline__ void store_acc(float* p, int64_t sb, int64_t sn, int b, int h, int row0,
                                          int nrows, int ct, int lq, int g, const f32x4& v) {
  const int c = 16 * ct + lq;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int node = row0 + 4 * g + r;
    if (node < nrows && c < DH) tok_row(p, sb, sn, b, node, h, DH)[c] = v[r];
  }
}

// cos(k pi / 2): T_k(0)
__device__ __forceinline__ float cheb_at_zero(int k) {
  return (k & 1) ? 0.0f : ((k & 2) ? -1.0f : 1.0f);
}

// ---- heads without a graph (Lhat = 0): y = X sum_k T_k(0) W_k + bias ---------------------

template <int DH>
__device__ __forceinline__ void nograph_fwd(const FilterArgs& a, int b, int h, int n, const float* w, int lq, int g) {
  constexpr int CT = Feat<DH>::CT;
  const int NTall = (a.N + 15) >> 4;
  for (int nt = 0; nt < NTall; ++nt) {
    f32x4 y[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) y[ct] = zero4();
    if (16 * nt < n) {
      const int node = 16 * nt + lq;
      Feat<DH> xf;
      load_row<DH>(xf, node < n ? tok_row(a.x, a.xsb, a.xsn, b, node, h, DH) : nullptr, g);
      for (int k = 0; k < a.P; k += 2) {
        f32x4 z[CT];
        xw_tile<DH>(xf, w, k, lq, g, z);
        const float sgn = cheb_at_zero(k);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) y[ct][r] += sgn * z[ct][r];
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c = 16 * ct + lq;
      const float bv = (a.bias != nullptr && c < DH) ? a.bias[c] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[ct][r] = (16 * nt + 4 * g + r < n) ? y[ct][r] + bv : 0.0f;
      store_acc<DH>(a.y, a.ysb, a.ysn, b, h, 16 * nt, a.N, ct, lq, g, y[ct]);
    }
  }
}

template <int DH>
__device__ void nograph_bwd(const FilterArgs& a, int b, int h, int n, const float* w, float* dw,
                            int item, int lq, int g) {
  constexpr int CT = Feat<DH>::CT;
  const int NTall = (a.N + 15) >> 4;
  const int NT = (n + 15) >> 4;
  f32x4 dbias[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) dbias[ct] = zero4();
  // dW_k = T_k(0) X^T dY : contraction over nodes (rows of both acc-layout tiles)
  for (int k = 0; k < a.P; ++k) {
    const float sgn = cheb_at_zero(k);
    f32x4 acc[CT][CT];
#pragma unroll
    for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
      for (int c2 = 0; c2 < CT; ++c2) acc[c1][c2] = zero4();
    if (sgn != 0.0f) {
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 xa[CT], dyb[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          xa[ct] = load_acc<DH>(a.x, a.xsb, a.xsn, b, h, 16 * nt, n, ct, lq, g);
          dyb[ct] = load_acc<DH>(a.dy, a.ysb, a.ysn, b, h, 16 * nt, n, ct, lq, g);
          if (k == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dbias[ct][r] += dyb[ct][r];
          }
        }
#pragma unroll
        for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
          for (int c2 = 0; c2 < CT; ++c2)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c1][c2] = mfma16(sgn * xa[c1][r], dyb[c2][r], acc[c1][c2]);
      }
    }
#pragma unroll
    for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
      for (int c2 = 0; c2 < CT; ++c2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * c1 + 4 * g + r, cp = 16 * c2 + lq;
          if (c < DH && cp < DH) dw[(k * DH + c) * DH + cp] = acc[c1][c2][r];
        }
  }
  // dbias partial
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    float s = dbias[ct][0] + dbias[ct][1] + dbias[ct][2] + dbias[ct][3];
    s += shfl_xor(s, 16);
    s += shfl_xor(s, 32);
    const int c = 16 * ct + lq;
    if (g == 0 && c < DH) a.dbias_part[(int64_t)item * DH + c] = s;
  }
  // dX = dY (sum_k T_k(0) W_k)^T, zero on padded rows
  for (int nt = 0; nt < NTall; ++nt) {
    f32x4 dx[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dx[ct] = zero4();
    if (nt < NT) {
      const int node = 16 * nt + lq;
      Feat<DH> dyf;
      load_row<DH>(dyf, node < n ? tok_row(a.dy, a.ysb, a.ysn, b, node, h, DH) : nullptr, g);
      for (int k = 0; k < a.P; k += 2) {
        f32x4 gk[CT];
        dyw_tile<DH>(dyf, w, k, lq, g, gk);
        const float sgn = cheb_at_zero(k);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) dx[ct][r] += sgn * gk[ct][r];
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) store_acc<DH>(a.dx, a.xsb, a.xsn, b, h, 16 * nt, a.N, ct, lq, g, dx[ct]);
  }
}

// ---- Chebyshev recursion on dense Lhat --------------------------------------------------

// out[tt] (+)= M . in over node tiles; M = Lhat (TRANS = false) or Lhat^T (TRANS = true).
// A operand: M[t = 16tt+lq][s = 16st+4g+r]; B operand: in[st][ct][r].
template <int DH, int NT_MAX, bool TRANS>
__device__ __forceinline__ void lhat_apply(const float* L, int N, int n, int NT, int lq, int g,
                                           const f32x4 (&in)[NT_MAX][Feat<DH>::CT],
                                           f32x4 (&out)[NT_MAX][Feat<DH>::CT]) {
  constexpr int CT = Feat<DH>::CT;
#pragma unroll
  for (int tt = 0; tt < NT_MAX; ++tt) {
    if (tt < NT) {
      f32x4 acc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = zero4();
      const int t = 16 * tt + lq;
#pragma unroll
      for (int st = 0; st < NT_MAX; ++st) {
        if (st < NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int s = 16 * st + 4 * g + r;
            float m = 0.0f;
            if (t < n && s < n) m = TRANS ? L[(int64_t)s * N + t] : L[(int64_t)t * N + s];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma16(m, in[st][ct][r], acc[ct]);
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) out[tt][ct] = acc[ct];
    }
  }
}

template <int DH, int NT_MAX>
__global__ __launch_bounds__(64 * kFWaves) void cheb_fwd_kernel(FilterArgs a) {
  constexpr int CT = Feat<DH>::CT;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kFWaves + wave_id();
  if (item >= a.total) return;
  const int h = item % a.H, b = item / a.H;
  const int n = a.n_real[b];
  const float* w = a.coeff + ((int64_t)h * a.B + b) * a.P * DH * DH;
  if (!a.share && h > 0) {
    nograph_fwd<DH>(a, b, h, n, w, lq, g);
    return;
  }
  const int NT = (n + 15) >> 4;
  const float* L = a.lhat + (int64_t)b * a.N * a.N;

  // Clenshaw: b_k = Z_k + 2 Lhat b_{k+1} - b_{k+2};  Y = Z_0 + Lhat b_1 - b_2
  f32x4 b1[NT_MAX][CT], b2[NT_MAX][CT], lb[NT_MAX][CT];
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      b1[nt][ct] = zero4();
      b2[nt][ct] = zero4();
      lb[nt][ct] = zero4();
    }
  for (int k = a.P - 1; k >= 0; --k) {
    if (k < a.P - 1) lhat_apply<DH, NT_MAX, false>(L, a.N, n, NT, lq, g, b1, lb);
    const float two = k == 0 ? 1.0f : 2.0f;
#pragma unroll
    for (int nt = 0; nt < NT_MAX; ++nt) {
      if (nt < NT) {
        const int node = 16 * nt + lq;
        Feat<DH> xf;
        load_row<DH>(xf, node < n ? tok_row(a.x, a.xsb, a.xsn, b, node, h, DH) : nullptr, g);
        f32x4 z[CT];
        xw_tile<DH>(xf, w, k, lq, g, z);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          f32x4 nb;
#pragma unroll
          for (int r = 0; r < 4; ++r) nb[r] = z[ct][r] + two * lb[nt][ct][r] - b2[nt][ct][r];
          b2[nt][ct] = b1[nt][ct];
          b1[nt][ct] = nb;
        }
      }
    }
  }
  // b1 now holds Y
  const int NTall = (a.N + 15) >> 4;
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    if (nt < NTall) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = 16 * ct + lq;
        const float bv = (a.bias != nullptr && c < DH) ? a.bias[c] : 0.0f;
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (nt < NT && 16 * nt + 4 * g + r < n) ? b1[nt][ct][r] + bv : 0.0f;
        store_acc<DH>(a.y, a.ysb, a.ysn, b, h, 16 * nt, a.N, ct, lq, g, y);
      }
    }
  }
}

template <int DH, int NT_MAX>
__global__ __launch_bounds__(64 * kFWaves) void cheb_bwd_kernel(FilterArgs a) {
  constexpr int CT = Feat<DH>::CT;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kFWaves + wave_id();
  if (item >= a.total) return;
  const int h = item % a.H, b = item / a.H;
  const int n = a.n_real[b];
  const int64_t blk = (int64_t)h * a.B + b;
  const float* w = a.coeff + blk * a.P * DH * DH;
  float* dw = a.dcoeff + blk * a.P * DH * DH;
  if (!a.share && h > 0) {
    nograph_bwd<DH>(a, b, h, n, w, dw, item, lq, g);
    return;
  }
  const int NT = (n + 15) >> 4;
  const int NTall = (a.N + 15) >> 4;
  const float* L = a.lhat + (int64_t)b * a.N * a.N;

  // phase 1: dW_k = (T_k(Lhat) X)^T dY, T_k by the forward recursion; dbias partial
  {
    f32x4 t0[NT_MAX][CT], t1[NT_MAX][CT], lt[NT_MAX][CT], dyb[NT_MAX][CT];
    f32x4 dbias[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dbias[ct] = zero4();
#pragma unroll
    for (int nt = 0; nt < NT_MAX; ++nt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        if (nt < NT) {
          t1[nt][ct] = load_acc<DH>(a.x, a.xsb, a.xsn, b, h, 16 * nt, n, ct, lq, g);
          dyb[nt][ct] = load_acc<DH>(a.dy, a.ysb, a.ysn, b, h, 16 * nt, n, ct, lq, g);
#pragma unroll
          for (int r = 0; r < 4; ++r) dbias[ct][r] += dyb[nt][ct][r];
        } else {
          t1[nt][ct] = zero4();
          dyb[nt][ct] = zero4();
        }
        t0[nt][ct] = zero4();
        lt[nt][ct] = zero4();
      }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float s = dbias[ct][0] + dbias[ct][1] + dbias[ct][2] + dbias[ct][3];
      s += shfl_xor(s, 16);
      s += shfl_xor(s, 32);
      const int c = 16 * ct + lq;
      if (g == 0 && c < DH) a.dbias_part[(int64_t)item * DH + c] = s;
    }
    // invariant at loop head: t1 = T_k, t0 = T_{k-1}
    for (int k = 0; k < a.P; ++k) {
      if (k >= 1) {
        lhat_apply<DH, NT_MAX, false>(L, a.N, n, NT, lq, g, t1, lt);
        const float two = k == 1 ? 1.0f : 2.0f;
#pragma unroll
        for (int nt = 0; nt < NT_MAX; ++nt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            f32x4 nw;
#pragma unroll
            for (int r = 0; r < 4; ++r) nw[r] = two * lt[nt][ct][r] - (k == 1 ? 0.0f : t0[nt][ct][r]);
            t0[nt][ct] = t1[nt][ct];
            t1[nt][ct] = nw;
          }
      }
      f32x4 acc[CT][CT];
#pragma unroll
      for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
        for (int c2 = 0; c2 < CT; ++c2) acc[c1][c2] = zero4();
#pragma unroll
      for (int nt = 0; nt < NT_MAX; ++nt) {
        if (nt < NT) {
#pragma unroll
          for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
            for (int c2 = 0; c2 < CT; ++c2)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[c1][c2] = mfma16(t1[nt][c1][r], dyb[nt][c2][r], acc[c1][c2]);
        }
      }
#pragma unroll
      for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
        for (int c2 = 0; c2 < CT; ++c2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * c1 + 4 * g + r, cp = 16 * c2 + lq;
            if (c < DH && cp < DH) dw[(k * DH + c) * DH + cp] = acc[c1][c2][r];
          }
    }
  }

  // phase 2: dX = sum_k T_k(Lhat^T) (dY W_k^T) by Clenshaw
  {
    f32x4 b1[NT_MAX][CT], b2[NT_MAX][CT], lb[NT_MAX][CT];
#pragma unroll
    for (int nt = 0; nt < NT_MAX; ++nt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        b1[nt][ct] = zero4();
        b2[nt][ct] = zero4();
        lb[nt][ct] = zero4();
      }
    for (int k = a.P - 1; k >= 0; --k) {
      if (k < a.P - 1) lhat_apply<DH, NT_MAX, true>(L, a.N, n, NT, lq, g, b1, lb);
      const float two = k == 0 ? 1.0f : 2.0f;
#pragma unroll
      for (int nt = 0; nt < NT_MAX; ++nt) {
        if (nt < NT) {
          const int node = 16 * nt + lq;
          Feat<DH> dyf;
          load_row<DH>(dyf, node < n ? tok_row(a.dy, a.ysb, a.ysn, b, node, h, DH) : nullptr, g);
          f32x4 gk[CT];
          dyw_tile<DH>(dyf, w, k, lq, g, gk);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            f32x4 nb;
#pragma unroll
            for (int r = 0; r < 4; ++r) nb[r] = gk[ct][r] + two * lb[nt][ct][r] - b2[nt][ct][r];
            b2[nt][ct] = b1[nt][ct];
            b1[nt][ct] = nb;
          }
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT_MAX; ++nt) {
      if (nt < NTall) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (nt < NT && 16 * nt + 4 * g + r < n) ? b1[nt][ct][r] : 0.0f;
          store_acc<DH>(a.dx, a.xsb, a.xsn, b, h, 16 * nt, a.N, ct, lq, g, v);
        }
      }
    }
  }
}

// ---- eigenbasis form ---------------------------------------------------------------------

// t_k(lam) for k < P into t[0..P-1] (P <= 8)
constexpr int kMaxOrder = 8;
__device__ __forceinline__ void cheb_poly(float lam, int P, float (&t)[kMaxOrder]) {
  t[0] = 1.0f;
  t[1] = lam;
#pragma unroll
  for (int k = 2; k < kMaxOrder; ++k) t[k] = (k < P) ? 2.0f * lam * t[k - 1] - t[k - 2] : 0.0f;
}

// Both directions are evaluated in the eigenbasis: X is projected first (Xtil = U^T X costs
// N K dh multiply-adds once, not once per order k), the P weight products run on the K x dh
// projected block, and one product with U returns to the node basis.  Orientation of every
// intermediate is chosen so that the next product contracts over its accumulator rows:
//   forward   Xtil^T [c][e]  --A-->  Ytil [e][c']  --B-->  Y [node][c']
//   backward  Xtil [e][c], dYtil [e][c']  --A,B-->  dW_k [c][c']
//             dYtil^T [c'][e]  --A-->  dXtil [e][c]  --B-->  dX [node][c]
// ("--A-->": the accumulator is fed back as the MFMA A operand, i <-> its column, k <-> its rows.)

template <int DH, int ET_MAX>
__global__ __launch_bounds__(64 * kFWaves) void spec_fwd_kernel(FilterArgs a) {
  constexpr int CT = Feat<DH>::CT;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kFWaves + wave_id();
  if (item >= a.total) return;
  const int h = item % a.H, b = item / a.H;
  const int n = a.n_real[b];
  const float* w = a.coeff + ((int64_t)h * a.B + b) * a.P * DH * DH;
  if (!a.share && h > 0) {
    nograph_fwd<DH>(a, b, h, n, w, lq, g);
    return;
  }
  const int NT = (n + 15) >> 4;
  const int NTall = (a.N + 15) >> 4;
  const int ET = (a.K + 15) >> 4;
  const float* U = a.u + (int64_t)b * a.N * a.K;
  const float* lam = a.lam + (int64_t)b * a.K;

  // (1) Xtil^T[c][e] = sum_node X[node][c] U[node][e]
  f32x4 xtT[CT][ET_MAX];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int et = 0; et < ET_MAX; ++et) xtT[ct][et] = zero4();
  for (int nt = 0; nt < NT; ++nt) {
    float xa[4][CT], uu[4][ET_MAX];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nd = 16 * nt + 4 * g + r;
      const int ndc = min(nd, n - 1);
      const float* row = tok_row(a.x, a.xsb, a.xsn, b, ndc, h, DH);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = 16 * ct + lq;
        const float v = row[c < DH ? c : 0];
        xa[r][ct] = (nd < n && c < DH) ? v : 0.0f;
      }
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        const int e = 16 * et + lq;
        const float v = U[(int64_t)ndc * a.K + min(e, a.K - 1)];
        uu[r][et] = (nd < n && e < a.K) ? v : 0.0f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et)
        if (et < ET) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) xtT[ct][et] = mfma16(xa[r][ct], uu[r][et], xtT[ct][et]);
        }
  }
  // (2) Ytil[e][c'] = sum_k t_k(lam_e) sum_c Xtil[e][c] W_k[c][c']
  f32x4 yt[ET_MAX][CT];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) yt[et][ct] = zero4();
    if (et < ET) {
      const int e = 16 * et + lq;
      float tk[kMaxOrder];
      cheb_poly(e < a.K ? lam[e] : 0.0f, a.P, tk);
#pragma unroll
      for (int k = 0; k < kMaxOrder; ++k) {
        if (k < a.P) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float av = xtT[ct][et][r] * tk[k];
#pragma unroll
              for (int c2 = 0; c2 < CT; ++c2)
                yt[et][c2] = mfma16(av, w_b<DH>(w, k, ct, r, c2, lq, g), yt[et][c2]);
            }
        }
      }
    }
  }
  // (3) Y = U Ytil + bias
  for (int nt = 0; nt < NTall; ++nt) {
    f32x4 y[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) y[ct] = zero4();
    if (nt < NT) {
      const int node = 16 * nt + lq;
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        if (et < ET) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = 16 * et + 4 * g + r;
            const float ua = (node < n && e < a.K) ? U[(int64_t)node * a.K + e] : 0.0f;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) y[ct] = mfma16(ua, yt[et][ct][r], y[ct]);
          }
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c = 16 * ct + lq;
      const float bv = (a.bias != nullptr && c < DH) ? a.bias[c] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[ct][r] = (16 * nt + 4 * g + r < n) ? y[ct][r] + bv : 0.0f;
      store_acc<DH>(a.y, a.ysb, a.ysn, b, h, 16 * nt, a.N, ct, lq, g, y[ct]);
    }
  }
}

template <int DH, int ET_MAX>
__global__ __launch_bounds__(64 * kFWaves) void spec_bwd_kernel(FilterArgs a) {
  constexpr int CT = Feat<DH>::CT;
  // dYtil^T doubles the live accumulators of pass (1); for wide shapes it gets its own pass over
  // the nodes (operands come from L2) instead of spilling
  constexpr bool kTwoPass = ET_MAX * CT > 8;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kFWaves + wave_id();
  if (item >= a.total) return;
  const int h = item % a.H, b = item / a.H;
  const int n = a.n_real[b];
  const int64_t blk = (int64_t)h * a.B + b;
  const float* w = a.coeff + blk * a.P * DH * DH;
  float* dw = a.dcoeff + blk * a.P * DH * DH;
  if (!a.share && h > 0) {
    nograph_bwd<DH>(a, b, h, n, w, dw, item, lq, g);
    return;
  }
  const int NT = (n + 15) >> 4;
  const int NTall = (a.N + 15) >> 4;
  const int ET = (a.K + 15) >> 4;
  const float* U = a.u + (int64_t)b * a.N * a.K;
  const float* lam = a.lam + (int64_t)b * a.K;

  // (1) Xtil = U^T X, dYtil = U^T dY (acc layout [e][c]), dYtil^T ([c'][e]); dbias partial
  f32x4 xt[ET_MAX][CT], dyt[ET_MAX][CT], dytT[CT][ET_MAX];
  f32x4 dbias[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) dbias[ct] = zero4();
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      xt[et][ct] = zero4();
      dyt[et][ct] = zero4();
      dytT[ct][et] = zero4();
    }
  for (int nt = 0; nt < NT; ++nt) {
    f32x4 xb[CT], dyb[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      xb[ct] = load_acc<DH>(a.x, a.xsb, a.xsn, b, h, 16 * nt, n, ct, lq, g);
      dyb[ct] = load_acc<DH>(a.dy, a.ysb, a.ysn, b, h, 16 * nt, n, ct, lq, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) dbias[ct][r] += dyb[ct][r];
    }
#pragma unroll
    for (int et = 0; et < ET_MAX; ++et) {
      if (et < ET) {
        const int e = 16 * et + lq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nd = 16 * nt + 4 * g + r;
          const float ua = (nd < n && e < a.K) ? U[(int64_t)nd * a.K + e] : 0.0f;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            xt[et][ct] = mfma16(ua, xb[ct][r], xt[et][ct]);
            dyt[et][ct] = mfma16(ua, dyb[ct][r], dyt[et][ct]);
            if constexpr (!kTwoPass) dytT[ct][et] = mfma16(dyb[ct][r], ua, dytT[ct][et]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    float s = dbias[ct][0] + dbias[ct][1] + dbias[ct][2] + dbias[ct][3];
    s += shfl_xor(s, 16);
    s += shfl_xor(s, 32);
    const int c = 16 * ct + lq;
    if (g == 0 && c < DH) a.dbias_part[(int64_t)item * DH + c] = s;
  }

  // (2) dW_k[c][c'] = sum_e t_k(lam_e) Xtil[e][c] dYtil[e][c']
  for (int k = 0; k < a.P; ++k) {
    f32x4 acc[CT][CT];
#pragma unroll
    for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
      for (int c2 = 0; c2 < CT; ++c2) acc[c1][c2] = zero4();
#pragma unroll
    for (int et = 0; et < ET_MAX; ++et) {
      if (et < ET) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = 16 * et + 4 * g + r;
          float tk[kMaxOrder];
          cheb_poly(e < a.K ? lam[e] : 0.0f, a.P, tk);
          float tke = 0.0f;
#pragma unroll
          for (int kk = 0; kk < kMaxOrder; ++kk)
            if (kk == k) tke = tk[kk];
#pragma unroll
          for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
            for (int c2 = 0; c2 < CT; ++c2)
              acc[c1][c2] = mfma16(xt[et][c1][r] * tke, dyt[et][c2][r], acc[c1][c2]);
        }
      }
    }
#pragma unroll
    for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
      for (int c2 = 0; c2 < CT; ++c2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * c1 + 4 * g + r, cp = 16 * c2 + lq;
          if (c < DH && cp < DH) dw[(k * DH + c) * DH + cp] = acc[c1][c2][r];
        }
  }

  if constexpr (kTwoPass) {
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 dyb[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        dyb[ct] = load_acc<DH>(a.dy, a.ysb, a.ysn, b, h, 16 * nt, n, ct, lq, g);
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        if (et < ET) {
          const int e = 16 * et + lq;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int nd = 16 * nt + 4 * g + r;
            const float ua = (nd < n && e < a.K) ? U[(int64_t)nd * a.K + e] : 0.0f;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dytT[ct][et] = mfma16(dyb[ct][r], ua, dytT[ct][et]);
          }
        }
      }
    }
  }

  // (3) dXtil[e][c] = sum_k t_k(lam_e) sum_c' dYtil[e][c'] W_k[c][c']   (overwrites xt)
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) xt[et][ct] = zero4();
    if (et < ET) {
      const int e = 16 * et + lq;
      float tk[kMaxOrder];
      cheb_poly(e < a.K ? lam[e] : 0.0f, a.P, tk);
#pragma unroll
      for (int k = 0; k < kMaxOrder; ++k) {
        if (k < a.P) {
#pragma unroll
          for (int c2 = 0; c2 < CT; ++c2)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const float4 wv = w_row4<DH>(w, k, ct, c2, lq, g);
#pragma unroll
              for (int r = 0; r < 4; ++r)
                xt[et][ct] = mfma16(dytT[c2][et][r] * tk[k], f4(wv, r), xt[et][ct]);
            }
        }
      }
    }
  }

  // (4) dX = U dXtil (rows >= n_real come out zero: their U rows are masked)
  for (int nt = 0; nt < NTall; ++nt) {
    f32x4 dx[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dx[ct] = zero4();
    if (nt < NT) {
      const int node = 16 * nt + lq;
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        if (et < ET) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = 16 * et + 4 * g + r;
            const float ub = (node < n && e < a.K) ? U[(int64_t)node * a.K + e] : 0.0f;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dx[ct] = mfma16(ub, xt[et][ct][r], dx[ct]);
          }
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      store_acc<DH>(a.dx, a.xsb, a.xsn, b, h, 16 * nt, a.N, ct, lq, g, dx[ct]);
  }
}

// ---- eigenbasis form, small graphs (N <= 64, K <= 32, dh <= 16): batched loads --------------
// Same arithmetic as spec_fwd/bwd_kernel; every global operand of the (graph, head) item - W_k,
// the X / dY rows, both orientations of the U tiles and lambda - is requested up front with
// clamped indices (unconditional loads, selects afterwards), so a wave pays one memory latency
// instead of one per inner-loop step.

template <int DH, int NT_MAX, int ET_MAX>
__global__ __launch_bounds__(64 * kFWaves) void spec_fwd_dense_kernel(FilterArgs a) {
  static_assert(DH <= 16, "dense variant: one feature chunk, one column tile");
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kFWaves + wave_id();
  if (item >= a.total) return;
  const int h = item % a.H, b = item / a.H;
  const int n = a.n_real[b];
  const float* w = a.coeff + ((int64_t)h * a.B + b) * a.P * DH * DH;
  if (!a.share && h > 0) {
    nograph_fwd<DH>(a, b, h, n, w, lq, g);
    return;
  }
  const float* U = a.u + (int64_t)b * a.N * a.K;
  const float* lam = a.lam + (int64_t)b * a.K;
  const int lqc = lq < DH ? lq : 0;
  const int nm1 = max(n - 1, 0);

  // ---- load batch -----------------------------------------------------------------------------
  float wB[kMaxOrder][4];  // W_k[c = 4g+s][c' = lq]
#pragma unroll
  for (int k = 0; k < kMaxOrder; ++k)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 4 * g + s;
      float v = 0.0f;
      if (k < a.P) v = w[(k * DH + (c < DH ? c : 0)) * DH + lqc];
      wB[k][s] = (k < a.P && c < DH && lq < DH) ? v : 0.0f;
    }
  float xa[NT_MAX][4];  // X[node = 16nt+4g+r][c = lq]
  float ua[NT_MAX][4][ET_MAX], ub[NT_MAX][4][ET_MAX], lamq[ET_MAX];
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    const int node = 16 * nt + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nd = 16 * nt + 4 * g + r;
      const int ndc = min(nd, nm1);
      const float xv = tok_row(a.x, a.xsb, a.xsn, b, ndc, h, DH)[lqc];
      xa[nt][r] = (nd < n && lq < DH) ? xv : 0.0f;
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        const int e = 16 * et + lq;
        const float v1 = U[(int64_t)ndc * a.K + min(e, a.K - 1)];
        ua[nt][r][et] = (nd < n && e < a.K) ? v1 : 0.0f;
        const int e2 = 16 * et + 4 * g + r;
        const float v2 = U[(int64_t)min(node, nm1) * a.K + min(e2, a.K - 1)];
        ub[nt][r][et] = (node < n && e2 < a.K) ? v2 : 0.0f;
      }
    }
  }
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) lamq[et] = lam[min(16 * et + lq, a.K - 1)];
  const float bv = (a.bias != nullptr) ? a.bias[lqc] : 0.0f;

  // ---- (1) Xtil^T[c][e] = sum_node X[node][c] U[node][e] ----------------------------------------
  f32x4 xtT[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) xtT[et] = zero4();
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    if (16 * nt < n) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int et = 0; et < ET_MAX; ++et) xtT[et] = mfma16(xa[nt][r], ua[nt][r][et], xtT[et]);
    }
  }
  // ---- (2) Ytil[e][c'] = sum_k t_k(lam_e) sum_c Xtil[e][c] W_k[c][c'] ---------------------------
  f32x4 yt[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    float tk[kMaxOrder];
    cheb_poly(lamq[et], a.P, tk);
    yt[et] = zero4();
#pragma unroll
    for (int k = 0; k < kMaxOrder; ++k) {
      if (k < a.P) {
#pragma unroll
        for (int r = 0; r < 4; ++r) yt[et] = mfma16(xtT[et][r] * tk[k], wB[k][r], yt[et]);
      }
    }
  }
  // ---- (3) Y = U Ytil + bias -------------------------------------------------------------------
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    if (16 * nt < a.N) {
      f32x4 y = zero4();
      if (16 * nt < n) {
#pragma unroll
        for (int et = 0; et < ET_MAX; ++et)
#pragma unroll
          for (int r = 0; r < 4; ++r) y = mfma16(ub[nt][r][et], yt[et][r], y);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = (16 * nt + 4 * g + r < n) ? y[r] + bv : 0.0f;
      store_acc<DH>(a.y, a.ysb, a.ysn, b, h, 16 * nt, a.N, 0, lq, g, y);
    }
  }
}

template <int DH, int NT_MAX, int ET_MAX>
__global__ __launch_bounds__(64 * kFWaves) void spec_bwd_dense_kernel(FilterArgs a) {
  static_assert(DH <= 16, "dense variant: one feature chunk, one column tile");
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kFWaves + wave_id();
  if (item >= a.total) return;
  const int h = item % a.H, b = item / a.H;
  const int n = a.n_real[b];
  const int64_t blk = (int64_t)h * a.B + b;
  const float* w = a.coeff + blk * a.P * DH * DH;
  float* dw = a.dcoeff + blk * a.P * DH * DH;
  if (!a.share && h > 0) {
    nograph_bwd<DH>(a, b, h, n, w, dw, item, lq, g);
    return;
  }
  const float* U = a.u + (int64_t)b * a.N * a.K;
  const float* lam = a.lam + (int64_t)b * a.K;
  const int lqc = lq < DH ? lq : 0;
  const int nm1 = max(n - 1, 0);

  // ---- load batch -----------------------------------------------------------------------------
  f32x4 xb[NT_MAX], dyb[NT_MAX];
  float ua[NT_MAX][4][ET_MAX], ub[NT_MAX][4][ET_MAX], lamr[ET_MAX][4], lamq[ET_MAX];
  float4 wr[kMaxOrder];  // W_k[c = lq][c' = 4g .. 4g+3]
#pragma unroll
  for (int k = 0; k < kMaxOrder; ++k) {
    wr[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (k < a.P) {
      const float4 v = *reinterpret_cast<const float4*>(w + (k * DH + lqc) * DH + (4 * g < DH ? 4 * g : 0));
      if (lq < DH && 4 * g < DH) wr[k] = v;
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    const int node = 16 * nt + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nd = 16 * nt + 4 * g + r;
      const int ndc = min(nd, nm1);
      const float xv = tok_row(a.x, a.xsb, a.xsn, b, ndc, h, DH)[lqc];
      const float dv = tok_row(a.dy, a.ysb, a.ysn, b, ndc, h, DH)[lqc];
      const bool ok = nd < n && lq < DH;
      xb[nt][r] = ok ? xv : 0.0f;
      dyb[nt][r] = ok ? dv : 0.0f;
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        const int e = 16 * et + lq;
        const float v1 = U[(int64_t)ndc * a.K + min(e, a.K - 1)];
        ua[nt][r][et] = (nd < n && e < a.K) ? v1 : 0.0f;
        const int e2 = 16 * et + 4 * g + r;
        const float v2 = U[(int64_t)min(node, nm1) * a.K + min(e2, a.K - 1)];
        ub[nt][r][et] = (node < n && e2 < a.K) ? v2 : 0.0f;
      }
    }
  }
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    lamq[et] = lam[min(16 * et + lq, a.K - 1)];
#pragma unroll
    for (int r = 0; r < 4; ++r) lamr[et][r] = lam[min(16 * et + 4 * g + r, a.K - 1)];
  }

  // ---- (1) Xtil = U^T X, dYtil = U^T dY ([e][c]), dYtil^T ([c'][e]); dbias partial --------------
  f32x4 xt[ET_MAX], dyt[ET_MAX], dytT[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    xt[et] = zero4();
    dyt[et] = zero4();
    dytT[et] = zero4();
  }
  float dbs = 0.0f;
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    if (16 * nt < n) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dbs += dyb[nt][r];
#pragma unroll
        for (int et = 0; et < ET_MAX; ++et) {
          xt[et] = mfma16(ua[nt][r][et], xb[nt][r], xt[et]);
          dyt[et] = mfma16(ua[nt][r][et], dyb[nt][r], dyt[et]);
          dytT[et] = mfma16(dyb[nt][r], ua[nt][r][et], dytT[et]);
        }
      }
    }
  }
  dbs += shfl_xor(dbs, 16);
  dbs += shfl_xor(dbs, 32);
  if (g == 0 && lq < DH) a.dbias_part[(int64_t)item * DH + lq] = dbs;

  float tkr[ET_MAX][4][kMaxOrder], tkq[ET_MAX][kMaxOrder];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    cheb_poly(lamq[et], a.P, tkq[et]);
#pragma unroll
    for (int r = 0; r < 4; ++r) cheb_poly(lamr[et][r], a.P, tkr[et][r]);
  }

  // ---- (2) dW_k[c][c'] = sum_e t_k(lam_e) Xtil[e][c] dYtil[e][c'] --------------------------------
#pragma unroll
  for (int k = 0; k < kMaxOrder; ++k) {
    if (k < a.P) {
      f32x4 acc = zero4();
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma16(xt[et][r] * tkr[et][r][k], dyt[et][r], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 4 * g + r;
        if (c < DH && lq < DH) dw[(k * DH + c) * DH + lq] = acc[r];
      }
    }
  }

  // ---- (3) dXtil[e][c] = sum_k t_k(lam_e) sum_c' dYtil[e][c'] W_k[c][c'] -------------------------
  f32x4 dxt[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    dxt[et] = zero4();
#pragma unroll
    for (int k = 0; k < kMaxOrder; ++k) {
      if (k < a.P) {
        const float t = tkq[et][k];
        dxt[et] = mfma16(dytT[et][0] * t, wr[k].x, dxt[et]);
        dxt[et] = mfma16(dytT[et][1] * t, wr[k].y, dxt[et]);
        dxt[et] = mfma16(dytT[et][2] * t, wr[k].z, dxt[et]);
        dxt[et] = mfma16(dytT[et][3] * t, wr[k].w, dxt[et]);
      }
    }
  }

  // ---- (4) dX = U dXtil (rows >= n_real come out zero: their U rows are masked) ------------------
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    if (16 * nt < a.N) {
      f32x4 dx = zero4();
      if (16 * nt < n) {
#pragma unroll
        for (int et = 0; et < ET_MAX; ++et)
#pragma unroll
          for (int r = 0; r < 4; ++r) dx = mfma16(ub[nt][r][et], dxt[et][r], dx);
      }
      store_acc<DH>(a.dx, a.xsb, a.xsn, b, h, 16 * nt, a.N, 0, lq, g, dx);
    }
  }
}

template <int DH, int NT_MAX, int ET_MAX>
int launch_spec_dense(const FilterArgs& a, bool bwd, hipStream_t stream) {
  const dim3 grid((a.total + kFWaves - 1) / kFWaves), block(64 * kFWaves);
  if (bwd) {
    auto kern = spec_bwd_dense_kernel<DH, NT_MAX, ET_MAX>;
    hipLaunchKernelGGL(kern, grid, block, 0, stream, a);
  } else {
    auto kern = spec_fwd_dense_kernel<DH, NT_MAX, ET_MAX>;
    hipLaunchKernelGGL(kern, grid, block, 0, stream, a);
  }
  return check_launch(bwd ? "feta_spec_filter_bwd" : "feta_spec_filter_fwd");
}

// -> 1 if the dense variant was launched (rc holds its status), 0 if the shape is outside it
template <int DH>
bool try_spec_dense_dh(const FilterArgs& a, bool bwd, hipStream_t stream, int* rc) {
  const int nt = (a.N + 15) / 16, et = (a.K + 15) / 16;
  if (nt > 4 || et > 2) return false;
  if (nt <= 3 && et <= 1) *rc = launch_spec_dense<DH, 3, 1>(a, bwd, stream);
  else if (et <= 1) *rc = launch_spec_dense<DH, 4, 1>(a, bwd, stream);
  else if (nt <= 3) *rc = launch_spec_dense<DH, 3, 2>(a, bwd, stream);
  else *rc = launch_spec_dense<DH, 4, 2>(a, bwd, stream);
  return true;
}

// ---- eigenbasis form, one workgroup per graph (d = 64: 4 heads x dh 16, N <= 192, K <= 32, K % 4 = 0) --
// The 4 waves of a workgroup are the 4 heads of ONE graph, so everything the heads share is fetched
// from HBM once and with 16-byte accesses: the eigenvector tile U_b [N, K] and lambda_b (the per-head
// kernels above read U eight times per graph: 4 heads x 2 operand orientations), and the node rows of
// X / dY as whole 256-byte rows (all heads) instead of one 64-byte head slice per wave.  Staged in LDS
// with pitches chosen for conflict-free operand reads (row pitch = 4 mod 8 floats: the two 4-row
// groups of a half-wave land on disjoint banks); both MFMA operand orientations of U come from the
// same LDS tile.  Results go back through LDS so that the stores are whole rows too.
constexpr int kGraphXP = 64 + 4;  // pitch of the staged [node][4 x 16] token rows
constexpr int kGraphWP = 16 + 4;  // pitch of the staged W_k rows

template <int NT_MAX, int ET_MAX, int PP>
__global__ __launch_bounds__(256) void spec_fwd_graph_kernel(FilterArgs a) {
  constexpr int DH = 16, XP = kGraphXP, UP = 16 * ET_MAX + 4, WP = kGraphWP, NR = 16 * NT_MAX;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, lq = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  const int n = a.n_real[b], nm1 = max(n - 1, 0);
  float* Xs = feta_lds;               // [NR][XP]
  float* Us = Xs + NR * XP;           // [NR][UP], zero outside [n, K]
  float* lams = Us + NR * UP;         // [16 ET_MAX]
  float* Ws = lams + 16 * ET_MAX + h * (PP * DH * WP);  // this head's [P * DH][WP]
  const float* U = a.u + (int64_t)b * a.N * a.K;
  const float* w = a.coeff + ((int64_t)h * a.B + b) * PP * DH * DH;

  // ---- cooperative loads: all requests first, LDS writes afterwards -----------------------------
  float4 xv[NT_MAX];
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, node = idx >> 4, q = idx & 15;
    const float4 v = *reinterpret_cast<const float4*>(tok_row(a.x, a.xsb, a.xsn, b, min(node, nm1), 0, DH) + 4 * q);
    xv[i] = keep4(node < n, v);
  }
  constexpr int UQ = 4 * ET_MAX, UI = (NR * UQ + 255) / 256;
  float4 uv[UI];
#pragma unroll
  for (int i = 0; i < UI; ++i) {
    const int idx = tid + 256 * i, node = idx / UQ, e = 4 * (idx % UQ);
    const float4 v = *reinterpret_cast<const float4*>(U + (int64_t)min(node, nm1) * a.K + min(e, a.K - 4));
    uv[i] = keep4(node < n && e < a.K, v);
  }
  // this head's W_k: requested first and written to LDS first (with 12 row tiles in flight the compiler
  // would otherwise park these registers in private memory)
  {
    float4 wv[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) wv[i] = reinterpret_cast<const float4*>(w)[lane + 64 * i];
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const int idx = lane + 64 * i;
      *reinterpret_cast<float4*>(Ws + (idx >> 2) * WP + 4 * (idx & 3)) = wv[i];
    }
  }
  const float lv = a.lam[(int64_t)b * a.K + min(tid, a.K - 1)];
  const float bv = (a.bias != nullptr) ? a.bias[lq] : 0.0f;
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i;
    *reinterpret_cast<float4*>(Xs + (idx >> 4) * XP + 4 * (idx & 15)) = xv[i];
  }
#pragma unroll
  for (int i = 0; i < UI; ++i) {
    const int idx = tid + 256 * i;
    if (idx < NR * UQ) *reinterpret_cast<float4*>(Us + (idx / UQ) * UP + 4 * (idx % UQ)) = uv[i];
  }
  if (tid < 16 * ET_MAX) lams[tid] = tid < a.K ? lv : 0.0f;
  __syncthreads();

  // ---- (1) Xtil^T[c][e] = sum_node X[node][c] U[node][e] ----------------------------------------
  f32x4 xtT[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) xtT[et] = zero4();
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    if (16 * nt < n) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nd = 16 * nt + 4 * g + r;
        const float xa = Xs[nd * XP + DH * h + lq];
#pragma unroll
        for (int et = 0; et < ET_MAX; ++et) xtT[et] = mfma16(xa, Us[nd * UP + 16 * et + lq], xtT[et]);
      }
    }
  }
  // ---- (2) Ytil[e][c'] = sum_k t_k(lam_e) sum_c Xtil[e][c] W_k[c][c'] ---------------------------
  f32x4 yt[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    float tk[kMaxOrder];
    cheb_poly(lams[16 * et + lq], PP, tk);
    yt[et] = zero4();
#pragma unroll
    for (int k = 0; k < PP; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        yt[et] = mfma16(xtT[et][r] * tk[k], Ws[(k * DH + 4 * g + r) * WP + lq], yt[et]);
  }
  // ---- (3) Y = U Ytil + bias, written over the X tile --------------------------------------------
  f32x4 y[NT_MAX];
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    y[nt] = zero4();
    if (16 * nt < n) {
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        const float4 ub = *reinterpret_cast<const float4*>(Us + (16 * nt + lq) * UP + 16 * et + 4 * g);
        y[nt] = mfma16(ub.x, yt[et][0], y[nt]);
        y[nt] = mfma16(ub.y, yt[et][1], y[nt]);
        y[nt] = mfma16(ub.z, yt[et][2], y[nt]);
        y[nt] = mfma16(ub.w, yt[et][3], y[nt]);
      }
    }
  }
  __syncthreads();  // every wave has read its X operands
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nd = 16 * nt + 4 * g + r;
      Xs[nd * XP + DH * h + lq] = nd < n ? y[nt][r] + bv : 0.0f;
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, node = idx >> 4, q = idx & 15;
    if (node < a.N)
      *reinterpret_cast<float4*>(tok_row(a.y, a.ysb, a.ysn, b, node, 0, DH) + 4 * q) =
          *reinterpret_cast<const float4*>(Xs + node * XP + 4 * q);
  }
}

template <int NT_MAX, int ET_MAX, int PP>
__global__ __launch_bounds__(256) void spec_bwd_graph_kernel(FilterArgs a) {
  constexpr int DH = 16, XP = kGraphXP, UP = 16 * ET_MAX + 4, NR = 16 * NT_MAX;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, lq = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  const int item = b * a.H + h;
  const int n = a.n_real[b], nm1 = max(n - 1, 0);
  float* Xs = feta_lds;               // [NR][XP]
  float* Ds = Xs + NR * XP;           // [NR][XP]  dY
  float* Us = Ds + NR * XP;           // [NR][UP]
  float* lams = Us + NR * UP;         // [16 ET_MAX]
  const float* U = a.u + (int64_t)b * a.N * a.K;
  const int64_t blk = (int64_t)h * a.B + b;
  const float* w = a.coeff + blk * PP * DH * DH;
  float* dw = a.dcoeff + blk * PP * DH * DH;

  // ---- cooperative loads ----------------------------------------------------------------------
  float4 xv[NT_MAX], dv[NT_MAX];
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, node = idx >> 4, q = idx & 15;
    const float4 v1 = *reinterpret_cast<const float4*>(tok_row(a.x, a.xsb, a.xsn, b, min(node, nm1), 0, DH) + 4 * q);
    const float4 v2 = *reinterpret_cast<const float4*>(tok_row(a.dy, a.ysb, a.ysn, b, min(node, nm1), 0, DH) + 4 * q);
    xv[i] = keep4(node < n, v1);
    dv[i] = keep4(node < n, v2);
  }
  constexpr int UQ = 4 * ET_MAX, UI = (NR * UQ + 255) / 256;
  float4 uv[UI];
#pragma unroll
  for (int i = 0; i < UI; ++i) {
    const int idx = tid + 256 * i, node = idx / UQ, e = 4 * (idx % UQ);
    const float4 v = *reinterpret_cast<const float4*>(U + (int64_t)min(node, nm1) * a.K + min(e, a.K - 4));
    uv[i] = keep4(node < n && e < a.K, v);
  }
  float4 wr[PP];  // W_k[c = lq][c' = 4g .. 4g+3]
#pragma unroll
  for (int k = 0; k < PP; ++k) wr[k] = *reinterpret_cast<const float4*>(w + (k * DH + lq) * DH + 4 * g);
  const float lv = a.lam[(int64_t)b * a.K + min(tid, a.K - 1)];
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, off = (idx >> 4) * XP + 4 * (idx & 15);
    *reinterpret_cast<float4*>(Xs + off) = xv[i];
    *reinterpret_cast<float4*>(Ds + off) = dv[i];
  }
#pragma unroll
  for (int i = 0; i < UI; ++i) {
    const int idx = tid + 256 * i;
    if (idx < NR * UQ) *reinterpret_cast<float4*>(Us + (idx / UQ) * UP + 4 * (idx % UQ)) = uv[i];
  }
  if (tid < 16 * ET_MAX) lams[tid] = tid < a.K ? lv : 0.0f;
  __syncthreads();

  // ---- (1) Xtil = U^T X, dYtil = U^T dY ([e][c]), dYtil^T ([c'][e]); dbias partial --------------
  f32x4 xt[ET_MAX], dyt[ET_MAX], dytT[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    xt[et] = zero4();
    dyt[et] = zero4();
    dytT[et] = zero4();
  }
  float dbs = 0.0f;
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    if (16 * nt < n) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nd = 16 * nt + 4 * g + r;
        const float xb = Xs[nd * XP + DH * h + lq];
        const float db = Ds[nd * XP + DH * h + lq];
        dbs += db;
#pragma unroll
        for (int et = 0; et < ET_MAX; ++et) {
          const float ua = Us[nd * UP + 16 * et + lq];
          xt[et] = mfma16(ua, xb, xt[et]);
          dyt[et] = mfma16(ua, db, dyt[et]);
          dytT[et] = mfma16(db, ua, dytT[et]);
        }
      }
    }
  }
  dbs += shfl_xor(dbs, 16);
  dbs += shfl_xor(dbs, 32);
  if (g == 0) a.dbias_part[(int64_t)item * DH + lq] = dbs;

  // ---- (2) dW_k[c][c'] = sum_e t_k(lam_e) Xtil[e][c] dYtil[e][c'] --------------------------------
  f32x4 dwa[PP];
#pragma unroll
  for (int k = 0; k < PP; ++k) dwa[k] = zero4();
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    const float4 l4 = *reinterpret_cast<const float4*>(lams + 16 * et + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float tk[kMaxOrder];
      cheb_poly(f4(l4, r), PP, tk);
#pragma unroll
      for (int k = 0; k < PP; ++k) dwa[k] = mfma16(xt[et][r] * tk[k], dyt[et][r], dwa[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < PP; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) dw[(k * DH + 4 * g + r) * DH + lq] = dwa[k][r];

  // ---- (3) dXtil[e][c] = sum_k t_k(lam_e) sum_c' dYtil[e][c'] W_k[c][c'] -------------------------
  f32x4 dxt[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    float tk[kMaxOrder];
    cheb_poly(lams[16 * et + lq], PP, tk);
    dxt[et] = zero4();
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      dxt[et] = mfma16(dytT[et][0] * tk[k], wr[k].x, dxt[et]);
      dxt[et] = mfma16(dytT[et][1] * tk[k], wr[k].y, dxt[et]);
      dxt[et] = mfma16(dytT[et][2] * tk[k], wr[k].z, dxt[et]);
      dxt[et] = mfma16(dytT[et][3] * tk[k], wr[k].w, dxt[et]);
    }
  }

  // ---- (4) dX = U dXtil, written over the X tile (rows >= n_real come out zero) ------------------
  f32x4 dx[NT_MAX];
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    dx[nt] = zero4();
    if (16 * nt < n) {
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        const float4 ub = *reinterpret_cast<const float4*>(Us + (16 * nt + lq) * UP + 16 * et + 4 * g);
        dx[nt] = mfma16(ub.x, dxt[et][0], dx[nt]);
        dx[nt] = mfma16(ub.y, dxt[et][1], dx[nt]);
        dx[nt] = mfma16(ub.z, dxt[et][2], dx[nt]);
        dx[nt] = mfma16(ub.w, dxt[et][3], dx[nt]);
      }
    }
  }
  __syncthreads();  // every wave has read its X / dY operands
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) Xs[(16 * nt + 4 * g + r) * XP + DH * h + lq] = dx[nt][r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, node = idx >> 4, q = idx & 15;
    if (node < a.N)
      *reinterpret_cast<float4*>(tok_row(a.dx, a.xsb, a.xsn, b, node, 0, DH) + 4 * q) =
          *reinterpret_cast<const float4*>(Xs + node * XP + 4 * q);
  }
}

// ---- linear_cat folded into the per-graph eigenbasis filter (round 4) --------------------------------------------
// transformer/models.py:223-224: output = linear_cat(cat(output, allout_filtered)) - behind the last layer the stack
// output and the filtered per-head outputs of a graph meet again, row by row.  With W_cat = [Wa | Wb]:
//     out = xn Wa^T + filt Wb^T + b_cat,   filt = U Ytil + bias   =>   filt Wb^T = U (Ytil Wb^T) + bias Wb^T
// i.e. the filter half of linear_cat is a [K x 64] x [64 x 64] product in the EIGEN domain (K = 16 rows instead of the
// graph's N), and the stack half is the graph's own rows through a 64 x 64 product - both fit this kernel's workgroup.
// One launch (feta_rowlin_fwd_ex over the virtual operand [xn | filt]) and one read of filt less per step; filt itself
// is still written: the backward of linear_cat contracts it with dout (dW_cat).
// xn = the stack output seen through its last BatchNorm: fresh statistics are finalized here (first consumer, the
// fused_stack.StackTail contract: workgroup 0 publishes the parameter block and the running statistics), or a published
// block, or the rows as they are (LayerNorm stack).  Rows of PADDED nodes (n_real <= node < N) take part: filt is zero there.
struct CatArgs {
  const float* y2;       // [N*B rows][64] stack output, row(b, i) = b * y2sb + i * y2sn (elements)
  int64_t y2sb, y2sn;
  const float* y2_bn;    // published [4][64] block, or NULL
  const float* y2_stats; // fresh partial sums [Gx + 1][2][64], or NULL
  int Gx;
  const float* gamma; const float* beta;
  float* bn_out; float* rmean; float* rvar; int64_t* nbt;
  float momentum, eps;
  int M;                 // rows the statistics run over (N * B)
  const float* w_cat;    // [64][128]
  const float* b_cat;    // [64] or NULL
  float* out;            // [N*B rows][64], strides of y
};

constexpr int kCatD = 64;

template <int NT_MAX, int ET_MAX>
__host__ __device__ inline int spec_cat_fwd_lds_floats(int pp) {
  constexpr int NR = 16 * NT_MAX, UP = 16 * ET_MAX + 4;
  return 2 * NR * kGraphXP + NR * UP + 16 * ET_MAX + 4 * pp * 16 * kGraphWP + 16 * ET_MAX * (kCatD + 4) + 2 * kCatD +
         reduce_scratch_floats(kCatD, 256);
}

template <int NT_MAX, int ET_MAX, int PP>
__global__ __launch_bounds__(256) void spec_cat_fwd_graph_kernel(FilterArgs a, CatArgs c) {
  constexpr int DH = 16, XP = kGraphXP, UP = 16 * ET_MAX + 4, WP = kGraphWP, NR = 16 * NT_MAX, D = kCatD, YP = D + 4;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, lq = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  const int n = a.n_real[b], nm1 = max(n - 1, 0), Nm1 = a.N - 1;
  float* Xs = feta_lds;               // [NR][XP] x rows (per-head outputs), later filt
  float* Ys = Xs + NR * XP;           // [NR][XP] stack output rows, normalised; later out
  float* Us = Ys + NR * XP;           // [NR][UP]
  float* lams = Us + NR * UP;         // [16 ET_MAX]
  float* Wsb = lams + 16 * ET_MAX;
  float* Ws = Wsb + h * (PP * DH * WP);   // this head's [P * DH][WP]
  float* YT = Wsb + 4 * PP * DH * WP;     // [16 ET_MAX][YP] Ytil of all heads
  float* xss = YT + 16 * ET_MAX * YP;     // [2][64] scale | shift of the BatchNorm in front
  float* scr = xss + 2 * D;               // finalize scratch
  const float* U = a.u + (int64_t)b * a.N * a.K;
  const float* w = a.coeff + ((int64_t)h * a.B + b) * PP * DH * DH;

  // ---- requests: everything this workgroup reads, before the first value is consumed ---------------------------------
  PartialBatchT<32> pb;
  partials_request_t<256, 32>(c.y2_stats != nullptr ? c.y2_stats : c.w_cat, c.y2_stats != nullptr ? c.Gx : 0, D, pb);
  float4 xv[NT_MAX], yv[NT_MAX];
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, node = idx >> 4, q = idx & 15;
    const float4 v = *reinterpret_cast<const float4*>(tok_row(a.x, a.xsb, a.xsn, b, min(node, nm1), 0, DH) + 4 * q);
    xv[i] = keep4(node < n, v);
    yv[i] = *reinterpret_cast<const float4*>(c.y2 + (int64_t)b * c.y2sb + (int64_t)min(node, Nm1) * c.y2sn + 4 * q);
  }
  constexpr int UQ = 4 * ET_MAX, UI = (NR * UQ + 255) / 256;
  float4 uv[UI];
#pragma unroll
  for (int i = 0; i < UI; ++i) {
    const int idx = tid + 256 * i, node = idx / UQ, e = 4 * (idx % UQ);
    const float4 v = *reinterpret_cast<const float4*>(U + (int64_t)min(node, nm1) * a.K + min(e, a.K - 4));
    uv[i] = keep4(node < n && e < a.K, v);
  }
  {
    float4 wv[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) wv[i] = reinterpret_cast<const float4*>(w)[lane + 64 * i];
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const int idx = lane + 64 * i;
      *reinterpret_cast<float4*>(Ws + (idx >> 2) * WP + 4 * (idx & 3)) = wv[i];
    }
  }
  // this wave's 16 rows of W_cat (output columns 16 h + lq), both halves, as row operands (feta_tiles.h)
  Feat<D> waf, wbf;
  load_row<D>(waf, c.w_cat + (int64_t)(DH * h + lq) * 2 * D, g);
  load_row<D>(wbf, c.w_cat + (int64_t)(DH * h + lq) * 2 * D + D, g);
  const float bcat = c.b_cat != nullptr ? c.b_cat[DH * h + lq] : 0.0f;
  const float lv = a.lam[(int64_t)b * a.K + min(tid, a.K - 1)];
  const float bv = (a.bias != nullptr) ? a.bias[lq] : 0.0f;
  float4 bs4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (a.bias != nullptr) bs4 = *reinterpret_cast<const float4*>(a.bias + 4 * g);
  float xg = 1.0f, xb = 0.0f, xk = 0.0f;
  if (c.y2_stats != nullptr && tid < D) {
    xg = c.gamma[tid];
    xb = c.beta[tid];
    xk = partials_shift(c.y2_stats, c.Gx, D, tid);
  }
  // ---- the BatchNorm in front of linear_cat -----------------------------------------------------------------------------
  if (c.y2_stats != nullptr) {
    reduce_partials_finish_t<256, 32>(c.y2_stats, c.Gx, D, pb, scr + 2 * D, scr);
    if (tid < D) {
      float mean, var;
      bn_moments_k(xk, D, c.M, scr, tid, mean, var);
      const float rstd = rsqrtf(var + c.eps);
      const float scale = xg * rstd, shift = xb - mean * scale;
      xss[tid] = scale;
      xss[D + tid] = shift;
      if (blockIdx.x == 0) {
        c.bn_out[tid] = scale;
        c.bn_out[D + tid] = shift;
        c.bn_out[2 * D + tid] = mean;
        c.bn_out[3 * D + tid] = rstd;
        if (c.rmean != nullptr) {
          const float unbiased = c.M > 1 ? var * (float)c.M / (float)(c.M - 1) : var;
          c.rmean[tid] = (1.0f - c.momentum) * c.rmean[tid] + c.momentum * mean;
          c.rvar[tid] = (1.0f - c.momentum) * c.rvar[tid] + c.momentum * unbiased;
        }
        if (tid == 0 && c.nbt != nullptr) *c.nbt += 1;
      }
    }
  } else if (tid < 2 * D) {
    xss[tid] = c.y2_bn != nullptr ? c.y2_bn[tid] : (tid < D ? 1.0f : 0.0f);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, q = idx & 15;
    *reinterpret_cast<float4*>(Xs + (idx >> 4) * XP + 4 * q) = xv[i];
    const float4 sc = *reinterpret_cast<const float4*>(xss + 4 * q), sh = *reinterpret_cast<const float4*>(xss + D + 4 * q);
    *reinterpret_cast<float4*>(Ys + (idx >> 4) * XP + 4 * q) =
        make_float4(yv[i].x * sc.x + sh.x, yv[i].y * sc.y + sh.y, yv[i].z * sc.z + sh.z, yv[i].w * sc.w + sh.w);
  }
#pragma unroll
  for (int i = 0; i < UI; ++i) {
    const int idx = tid + 256 * i;
    if (idx < NR * UQ) *reinterpret_cast<float4*>(Us + (idx / UQ) * UP + 4 * (idx % UQ)) = uv[i];
  }
  if (tid < 16 * ET_MAX) lams[tid] = tid < a.K ? lv : 0.0f;
  __syncthreads();

  // ---- (1) Xtil^T[c][e] = sum_node X[node][c] U[node][e] ----------------------------------------
  f32x4 xtT[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) xtT[et] = zero4();
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    if (16 * nt < n) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nd = 16 * nt + 4 * g + r;
        const float xa = Xs[nd * XP + DH * h + lq];
#pragma unroll
        for (int et = 0; et < ET_MAX; ++et) xtT[et] = mfma16(xa, Us[nd * UP + 16 * et + lq], xtT[et]);
      }
    }
  }
  // ---- (2) Ytil[e][c'] = sum_k t_k(lam_e) sum_c Xtil[e][c] W_k[c][c'] ---------------------------
  f32x4 yt[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    float tk[kMaxOrder];
    cheb_poly(lams[16 * et + lq], PP, tk);
    yt[et] = zero4();
#pragma unroll
    for (int k = 0; k < PP; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        yt[et] = mfma16(xtT[et][r] * tk[k], Ws[(k * DH + 4 * g + r) * WP + lq], yt[et]);
    // every head's Ytil meets in LDS: the operand of the eigen-domain half of linear_cat
#pragma unroll
    for (int r = 0; r < 4; ++r) YT[(16 * et + 4 * g + r) * YP + DH * h + lq] = yt[et][r];
  }
  __syncthreads();
  // ---- (2') YtilP[e][o] = sum_c Ytil[e][c] Wb[o][c], o = 16 h + lq: (row e = 4g + r, column o = lq) ---------------
  f32x4 ytp[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    Feat<D> yf;
    load_row<D>(yf, YT + (16 * et + lq) * YP, g);
    ytp[et] = dot_rows<D>(yf, wbf, zero4());
  }
  // (bias Wb^T)[o]: the filter's bias is one [16] vector for every head; this lane holds Wb[o][16 j + 4 g + s]
  float cbo = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    cbo += (wbf.f[j][0] * bs4.x + wbf.f[j][1] * bs4.y) + (wbf.f[j][2] * bs4.z + wbf.f[j][3] * bs4.w);
  cbo += shfl_xor(cbo, 16);
  cbo += shfl_xor(cbo, 32);
  // ---- (3) filt = U Ytil + bias;  out = xn Wa^T + U YtilP + bias Wb^T + b_cat --------------------------------------
  f32x4 y[NT_MAX], o[NT_MAX];
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    y[nt] = zero4();
    o[nt] = zero4();
    if (16 * nt < a.N) {
      Feat<D> xf;
      load_row<D>(xf, Ys + (16 * nt + lq) * XP, g);
      o[nt] = dot_rows<D>(xf, waf, zero4());   // (row node = 4g + r, column o = lq)
    }
    if (16 * nt < n) {
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        const float4 ub = *reinterpret_cast<const float4*>(Us + (16 * nt + lq) * UP + 16 * et + 4 * g);
        y[nt] = mfma16(ub.x, yt[et][0], y[nt]);
        y[nt] = mfma16(ub.y, yt[et][1], y[nt]);
        y[nt] = mfma16(ub.z, yt[et][2], y[nt]);
        y[nt] = mfma16(ub.w, yt[et][3], y[nt]);
        o[nt] = mfma16(ub.x, ytp[et][0], o[nt]);
        o[nt] = mfma16(ub.y, ytp[et][1], o[nt]);
        o[nt] = mfma16(ub.z, ytp[et][2], o[nt]);
        o[nt] = mfma16(ub.w, ytp[et][3], o[nt]);
      }
    }
  }
  __syncthreads();  // every wave has read its X / Y operands
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nd = 16 * nt + 4 * g + r;
      Xs[nd * XP + DH * h + lq] = nd < n ? y[nt][r] + bv : 0.0f;
      Ys[nd * XP + DH * h + lq] = o[nt][r] + bcat + (nd < n ? cbo : 0.0f);
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, node = idx >> 4, q = idx & 15;
    if (node < a.N) {
      *reinterpret_cast<float4*>(tok_row(a.y, a.ysb, a.ysn, b, node, 0, DH) + 4 * q) =
          *reinterpret_cast<const float4*>(Xs + node * XP + 4 * q);
      *reinterpret_cast<float4*>(tok_row(c.out, a.ysb, a.ysn, b, node, 0, DH) + 4 * q) =
          *reinterpret_cast<const float4*>(Ys + node * XP + 4 * q);
    }
  }
}

template <int NT_MAX, int ET_MAX>
int launch_spec_cat_fwd(const FilterArgs& a, const CatArgs& c, hipStream_t stream) {
  const size_t lds = sizeof(float) * spec_cat_fwd_lds_floats<NT_MAX, ET_MAX>(4);
  auto kern = spec_cat_fwd_graph_kernel<NT_MAX, ET_MAX, 4>;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), lds, stream, a, c);
  return check_launch("feta_spec_filter_cat_fwd");
}

// ---- backward of the filter stage with linear_cat folded in (round 4, ABI 11) ------------------------------------------
// What feta_rowlin_bwd_ex (linear_cat: dx_n, dfilt, dW_cat, db_cat, BatchNorm-backward sums) and feta_spec_filter_bwd did in
// two launches, per graph in one.  dout = the gradient of linear_cat's output, W_cat = [Wa | Wb]:
//     dfilt = dout Wb  =>  U^T dfilt = (U^T dout) Wb = Dt Wb           (a [K x 64] x [64 x 64] product, never N rows)
//     dbias_filter     = (sum_{node < n} dout) Wb
//     dW_b = sum_node dout^T filt,  dW_a = sum_node dout^T xn,  db_cat = sum_node dout      (per-graph partial row)
//     dxn  = dout Wa,  gs = (sum dxn, sum dxn xhat)                     (fused_stack.StackTail contract, one row per graph)
// Wave h owns columns 16 h .. 16 h + 15 of every 64-wide quantity (its head's columns of filt / x, its 16 outputs of Dt).
// Rows of padded nodes (n_real <= node < N) take part in everything but the eigen-domain terms: filt is zero there, U too.
constexpr int kCatBwdMaxGrid = 512;   // workgroups of a backward launch = rows of its partial / gs outputs
inline int spec_cat_bwd_rows(int B) { return B < kCatBwdMaxGrid ? B : kCatBwdMaxGrid; }

struct CatGradArgs {
  const float* dout;     // [rows][64], row(b, i) = b * y2sb + i * y2sn (the strides of y2: both are [N, B, 64] tensors)
  const float* y2;       // stack output rows (pre-norm if y2_bn)
  int64_t y2sb, y2sn;
  const float* y2_bn;    // published [4][64] block (scale, shift, mean, rstd) or NULL: xn = y2
  const float* filt;     // the forward's filter output (strides ysb, ysn of the call)
  const float* w_cat;    // [64][128]
  float* dxn;            // [rows][64], strides of y2
  float* gs;             // [B][2][64] or NULL
  float* partial;        // [B][partial_ld]: dW_cat [64][128] | db_cat [64]
  int64_t partial_ld;
};

template <int NT_MAX, int ET_MAX>
__host__ __device__ inline int spec_cat_bwd_lds_floats() {
  constexpr int NR = 16 * NT_MAX, UP = 16 * ET_MAX + 4;
  return 2 * NR * kGraphXP + NR * UP + 16 * ET_MAX + 16 * ET_MAX * (kCatD + 4) + 2 * kCatD;
}

template <int NT_MAX, int ET_MAX, int PP>
__global__ __launch_bounds__(256) void spec_cat_bwd_graph_kernel(FilterArgs a, CatGradArgs c) {
  constexpr int DH = 16, XP = kGraphXP, UP = 16 * ET_MAX + 4, NR = 16 * NT_MAX, D = kCatD, DP = D + 4;
  const int h = threadIdx.x >> 6, lane0 = threadIdx.x & 63;
  const int Nm1 = a.N - 1;
  const int col = DH * h + (lane0 & 15);   // this lane's column of the 64-wide rows
  float* Xs = feta_lds;               // [NR][XP] x rows (per-head outputs), later dx
  float* Ds = Xs + NR * XP;           // [NR][XP] dout rows (all N of them), later dxn
  float* Us = Ds + NR * XP;           // [NR][UP]
  float* lams = Us + NR * UP;         // [16 ET_MAX]
  float* DT = lams + 16 * ET_MAX;     // [16 ET_MAX][DP]  Dt = U^T dout
  float* SR = DT + 16 * ET_MAX * DP;  // [2][64] column sums of dout: real nodes | all N rows
  // W_cat[o = 16 j + 4 g + s][this lane's column], both halves: the k-slot g of step (j, s) stands for output o
  float wa[16], wb[16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const float* row = c.w_cat + (int64_t)(16 * j + 4 * (lane0 >> 4) + s_) * 2 * D;
      wa[4 * j + s_] = row[col];
      wb[4 * j + s_] = row[D + col];
    }
  float bsc = 1.0f, bsh = 0.0f, bmean = 0.0f, brstd = 0.0f;
  if (c.y2_bn != nullptr) {
    bsc = c.y2_bn[col];
    bsh = c.y2_bn[D + col];
    bmean = c.y2_bn[2 * D + col];
    brstd = c.y2_bn[3 * D + col];
  }
  // what a workgroup leaves per LAUNCH, not per graph: its row of the dW_cat / db_cat partials and of the BatchNorm-backward
  // sums.  A batch beyond the grid (launch_spec_cat_bwd: 512 workgroups) is walked, and these stay in registers across the
  // graphs - one row per graph was 1024 rows of 33 KB at config 5 for the reduction launch and for feta_ffn_bwd's prologue
  f32x4 gaw[4], gbw[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    gaw[t] = zero4();
    gbw[t] = zero4();
  }
  float sall_w = 0.0f, g1 = 0.0f, g2 = 0.0f;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
  if (b != (int)blockIdx.x) __syncthreads();   // the previous graph's tiles have been written out
  // (what a lane derives from its id is invariant in the graph loop and would be hoisted and held across the whole body -
  // 256 + 111 registers; the lane id is laundered once per graph: csrc/block_bwd.hip)
  int lane_l = lane0;
  FETA_OPAQUE_LANE(lane_l);
  const int lane = lane_l, tid = (h << 6) | lane, lq = lane & 15, g = lane >> 4;
  const int item = b * a.H + h;
  const int n = a.n_real[b], nm1 = max(n - 1, 0);
  const float* U = a.u + (int64_t)b * a.N * a.K;
  const int64_t blk = (int64_t)h * a.B + b;
  const float* w = a.coeff + blk * PP * DH * DH;
  float* dw = a.dcoeff + blk * PP * DH * DH;
  const int64_t rb = (int64_t)b * c.y2sb;

  // ---- requests ------------------------------------------------------------------------------------------------------
  float4 xv[NT_MAX], dv[NT_MAX];
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, node = idx >> 4, q = idx & 15;
    const float4 v1 = *reinterpret_cast<const float4*>(tok_row(a.x, a.xsb, a.xsn, b, min(node, nm1), 0, DH) + 4 * q);
    const float4 v2 = *reinterpret_cast<const float4*>(c.dout + rb + (int64_t)min(node, Nm1) * c.y2sn + 4 * q);
    xv[i] = keep4(node < n, v1);
    dv[i] = keep4(node < a.N, v2);
  }
  constexpr int UQ = 4 * ET_MAX, UI = (NR * UQ + 255) / 256;
  float4 uv[UI];
#pragma unroll
  for (int i = 0; i < UI; ++i) {
    const int idx = tid + 256 * i, node = idx / UQ, e = 4 * (idx % UQ);
    const float4 v = *reinterpret_cast<const float4*>(U + (int64_t)min(node, nm1) * a.K + min(e, a.K - 4));
    uv[i] = keep4(node < n && e < a.K, v);
  }
  float4 wr[PP];  // W_k[c = lq][c' = 4g .. 4g+3]
#pragma unroll
  for (int k = 0; k < PP; ++k) wr[k] = *reinterpret_cast<const float4*>(w + (k * DH + lq) * DH + 4 * g);
  const float lv = a.lam[(int64_t)b * a.K + min(tid, a.K - 1)];
  // this lane's column of the stack-output and filter-output rows node = 16 nt + 4 g + r (k-slot g of step (nt, r))
  float xn[NT_MAX][4], fl[NT_MAX][4], xh[NT_MAX][4];
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int node = 16 * nt + 4 * g + r, nc = min(node, Nm1);
      const float yv = c.y2[rb + (int64_t)nc * c.y2sn + col];
      const float fv = tok_row(c.filt, a.ysb, a.ysn, b, nc, h, DH)[lq];
      const bool ok = node < a.N;
      xn[nt][r] = ok ? yv * bsc + bsh : 0.0f;
      xh[nt][r] = ok ? (yv - bmean) * brstd : 0.0f;
      fl[nt][r] = ok ? fv : 0.0f;
    }
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, off = (idx >> 4) * XP + 4 * (idx & 15);
    *reinterpret_cast<float4*>(Xs + off) = xv[i];
    *reinterpret_cast<float4*>(Ds + off) = dv[i];
  }
#pragma unroll
  for (int i = 0; i < UI; ++i) {
    const int idx = tid + 256 * i;
    if (idx < NR * UQ) *reinterpret_cast<float4*>(Us + (idx / UQ) * UP + 4 * (idx % UQ)) = uv[i];
  }
  if (tid < 16 * ET_MAX) lams[tid] = tid < a.K ? lv : 0.0f;
  __syncthreads();

  // ---- (1) Xtil = U^T X; Dt[e][o = 16 h + lq] = sum_{node < n} U[node][e] dout[node][o]; column sums of dout ----------
  f32x4 xt[ET_MAX], dt[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    xt[et] = zero4();
    dt[et] = zero4();
  }
  float sreal = 0.0f, sall = 0.0f;
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    if (16 * nt < a.N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nd = 16 * nt + 4 * g + r;
        const float db = Ds[nd * XP + col];
        sall += db;
        if (16 * nt < n) {
          const float xb = Xs[nd * XP + col];
          sreal += nd < n ? db : 0.0f;
#pragma unroll
          for (int et = 0; et < ET_MAX; ++et) {
            const float ua = Us[nd * UP + 16 * et + lq];   // (zero for node >= n)
            xt[et] = mfma16(ua, xb, xt[et]);
            dt[et] = mfma16(ua, db, dt[et]);
          }
        }
      }
    }
  }
  sreal += shfl_xor(sreal, 16);
  sreal += shfl_xor(sreal, 32);
  sall += shfl_xor(sall, 16);
  sall += shfl_xor(sall, 32);
  if (g == 0) {
    SR[col] = sreal;
    SR[D + col] = sall;
  }
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et)
#pragma unroll
    for (int r = 0; r < 4; ++r) DT[(16 * et + 4 * g + r) * DP + col] = dt[et][r];
  __syncthreads();

  // ---- (1') dYtil = Dt Wb ([e][c'] and its transpose), dbias_filter = (sum_real dout) Wb --------------------------------
  f32x4 dyt[ET_MAX], dytT[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    dyt[et] = zero4();
    dytT[et] = zero4();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 d4 = *reinterpret_cast<const float4*>(DT + (16 * et + lq) * DP + 16 * j + 4 * g);   // Dt[e = lq][o]
      dyt[et] = mfma16(d4.x, wb[4 * j], dyt[et]);           // (row e = 4g + r, column c' = lq)
      dyt[et] = mfma16(d4.y, wb[4 * j + 1], dyt[et]);
      dyt[et] = mfma16(d4.z, wb[4 * j + 2], dyt[et]);
      dyt[et] = mfma16(d4.w, wb[4 * j + 3], dyt[et]);
      dytT[et] = mfma16(wb[4 * j], d4.x, dytT[et]);         // (row c' = 4g + r, column e = lq)
      dytT[et] = mfma16(wb[4 * j + 1], d4.y, dytT[et]);
      dytT[et] = mfma16(wb[4 * j + 2], d4.z, dytT[et]);
      dytT[et] = mfma16(wb[4 * j + 3], d4.w, dytT[et]);
    }
  }
  {
    float dbs = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 s4 = *reinterpret_cast<const float4*>(SR + 16 * j + 4 * g);
      dbs += (s4.x * wb[4 * j] + s4.y * wb[4 * j + 1]) + (s4.z * wb[4 * j + 2] + s4.w * wb[4 * j + 3]);
    }
    dbs += shfl_xor(dbs, 16);
    dbs += shfl_xor(dbs, 32);
    if (g == 0) a.dbias_part[(int64_t)item * DH + lq] = dbs;
  }

  // ---- (2) dW_k[c][c'] = sum_e t_k(lam_e) Xtil[e][c] dYtil[e][c'] --------------------------------
  f32x4 dwa[PP];
#pragma unroll
  for (int k = 0; k < PP; ++k) dwa[k] = zero4();
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    const float4 l4 = *reinterpret_cast<const float4*>(lams + 16 * et + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float tk[kMaxOrder];
      cheb_poly(f4(l4, r), PP, tk);
#pragma unroll
      for (int k = 0; k < PP; ++k) dwa[k] = mfma16(xt[et][r] * tk[k], dyt[et][r], dwa[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < PP; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) dw[(k * DH + 4 * g + r) * DH + lq] = dwa[k][r];

  // ---- (3) dXtil[e][c] = sum_k t_k(lam_e) sum_c' dYtil[e][c'] W_k[c][c'] -------------------------
  f32x4 dxt[ET_MAX];
#pragma unroll
  for (int et = 0; et < ET_MAX; ++et) {
    float tk[kMaxOrder];
    cheb_poly(lams[16 * et + lq], PP, tk);
    dxt[et] = zero4();
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      dxt[et] = mfma16(dytT[et][0] * tk[k], wr[k].x, dxt[et]);
      dxt[et] = mfma16(dytT[et][1] * tk[k], wr[k].y, dxt[et]);
      dxt[et] = mfma16(dytT[et][2] * tk[k], wr[k].z, dxt[et]);
      dxt[et] = mfma16(dytT[et][3] * tk[k], wr[k].w, dxt[et]);
    }
  }

  // ---- (4) dX = U dXtil (rows >= n_real come out zero) -------------------------------------------
  f32x4 dx[NT_MAX];
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    dx[nt] = zero4();
    if (16 * nt < n) {
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        const float4 ub = *reinterpret_cast<const float4*>(Us + (16 * nt + lq) * UP + 16 * et + 4 * g);
        dx[nt] = mfma16(ub.x, dxt[et][0], dx[nt]);
        dx[nt] = mfma16(ub.y, dxt[et][1], dx[nt]);
        dx[nt] = mfma16(ub.z, dxt[et][2], dx[nt]);
        dx[nt] = mfma16(ub.w, dxt[et][3], dx[nt]);
      }
    }
  }

  // ---- (5) linear_cat: dW_a / dW_b tiles (rows o = 16 t + 4g + r, this lane's column), dxn = dout Wa ----------------------
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int nt = 0; nt < NT_MAX; ++nt) {
      if (16 * nt < a.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float da = Ds[(16 * nt + 4 * g + r) * XP + 16 * t + lq];   // dout[node][o = 16 t + lq]
          gaw[t] = mfma16(da, xn[nt][r], gaw[t]);
          gbw[t] = mfma16(da, fl[nt][r], gbw[t]);
        }
      }
    }
  }
  sall_w += sall;
  f32x4 dn[NT_MAX];
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt) {
    dn[nt] = zero4();
    if (16 * nt < a.N) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 d4 = *reinterpret_cast<const float4*>(Ds + (16 * nt + lq) * XP + 16 * j + 4 * g);   // dout[node = lq][o]
        dn[nt] = mfma16(d4.x, wa[4 * j], dn[nt]);           // (row node = 4g + r, this lane's column)
        dn[nt] = mfma16(d4.y, wa[4 * j + 1], dn[nt]);
        dn[nt] = mfma16(d4.z, wa[4 * j + 2], dn[nt]);
        dn[nt] = mfma16(d4.w, wa[4 * j + 3], dn[nt]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = 16 * nt + 4 * g + r < a.N ? dn[nt][r] : 0.0f;
        g1 += v;
        g2 += v * xh[nt][r];
      }
    }
  }
  __syncthreads();  // every wave has read its X / dout operands
#pragma unroll
  for (int nt = 0; nt < NT_MAX; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Xs[(16 * nt + 4 * g + r) * XP + col] = dx[nt][r];
      Ds[(16 * nt + 4 * g + r) * XP + col] = dn[nt][r];
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NT_MAX; ++i) {
    const int idx = tid + 256 * i, node = idx >> 4, q = idx & 15;
    if (node < a.N) {
      *reinterpret_cast<float4*>(tok_row(a.dx, a.xsb, a.xsn, b, node, 0, DH) + 4 * q) =
          *reinterpret_cast<const float4*>(Xs + node * XP + 4 * q);
      *reinterpret_cast<float4*>(c.dxn + rb + (int64_t)node * c.y2sn + 4 * q) =
          *reinterpret_cast<const float4*>(Ds + node * XP + 4 * q);
    }
  }
  }  // graphs of this workgroup
  const int g = lane0 >> 4;
  float* prow = c.partial + (int64_t)blockIdx.x * c.partial_ld;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      prow[(int64_t)(16 * t + 4 * g + r) * 2 * D + col] = gaw[t][r];
      prow[(int64_t)(16 * t + 4 * g + r) * 2 * D + D + col] = gbw[t][r];
    }
  if (g == 0) prow[(int64_t)D * 2 * D + col] = sall_w;   // db_cat
  if (c.gs != nullptr) {
    g1 += shfl_xor(g1, 16);
    g1 += shfl_xor(g1, 32);
    g2 += shfl_xor(g2, 16);
    g2 += shfl_xor(g2, 32);
    if (g == 0) {
      c.gs[(int64_t)blockIdx.x * 2 * D + col] = g1;
      c.gs[(int64_t)blockIdx.x * 2 * D + D + col] = g2;
    }
  }
}

template <int NT_MAX, int ET_MAX>
int launch_spec_cat_bwd(const FilterArgs& a, const CatGradArgs& c, hipStream_t stream) {
  const size_t lds = sizeof(float) * spec_cat_bwd_lds_floats<NT_MAX, ET_MAX>();
  auto kern = spec_cat_bwd_graph_kernel<NT_MAX, ET_MAX, 4>;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  hipLaunchKernelGGL(kern, dim3(spec_cat_bwd_rows(a.B)), dim3(256), lds, stream, a, c);
  return check_launch("feta_spec_filter_cat_bwd");
}

template <int NT_MAX, int ET_MAX, int PP>
int launch_spec_graph_p(const FilterArgs& a, bool bwd, hipStream_t stream) {
  constexpr int NR = 16 * NT_MAX, UP = 16 * ET_MAX + 4;
  const dim3 grid(a.B), block(256);
  if (bwd) {
    const size_t lds = sizeof(float) * (2 * NR * kGraphXP + NR * UP + 16 * ET_MAX);
    auto kern = spec_bwd_graph_kernel<NT_MAX, ET_MAX, PP>;
    static LdsSeen lds_seen;
    allow_dynamic_lds(kern, lds, lds_seen);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  } else {
    const size_t lds = sizeof(float) * (NR * kGraphXP + NR * UP + 16 * ET_MAX + 4 * PP * 16 * kGraphWP);
    auto kern = spec_fwd_graph_kernel<NT_MAX, ET_MAX, PP>;
    static LdsSeen lds_seen;
    allow_dynamic_lds(kern, lds, lds_seen);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  }
  return check_launch(bwd ? "feta_spec_filter_bwd" : "feta_spec_filter_fwd");
}

template <int NT_MAX, int ET_MAX>
int launch_spec_graph(const FilterArgs& a, bool bwd, hipStream_t stream) {
  switch (a.P) {   // the filter order is a compile-time constant of these kernels (branch-free operand batches)
    case 2: return launch_spec_graph_p<NT_MAX, ET_MAX, 2>(a, bwd, stream);
    case 3: return launch_spec_graph_p<NT_MAX, ET_MAX, 3>(a, bwd, stream);
    case 4: return launch_spec_graph_p<NT_MAX, ET_MAX, 4>(a, bwd, stream);
    default: return launch_spec_graph_p<NT_MAX, ET_MAX, 5>(a, bwd, stream);
  }
}

// -> true if the one-workgroup-per-graph variant was launched.  It needs all heads on the graph
// (heads_share_graph), 4 heads x dh 16 stored as one 64-float row per node, 16-byte aligned rows of U.
bool try_spec_graph(const FilterArgs& a, int dh, bool bwd, hipStream_t stream, int* rc) {
  const int nt = (a.N + 15) / 16, et = (a.K + 15) / 16;
  if (!a.share || a.H != 4 || dh != 16 || nt > 12 || et > 2 || (a.K & 3) != 0 || a.P < 2 || a.P > 5) return false;
  if (!aligned16(a.u)) return false;
  if (nt > 4) {
    // large graphs (N <= 192, the PATTERN shape): the node rows of X / dY and the eigenvector tile U_b still
    // fit in LDS (100 KB forward, 132 KB backward) - the "LDS-tiled U^T X" of BASELINE config 4.  Only the
    // default filter order is instantiated at these sizes; other orders take the per-head kernels.
    if (a.P != 4) return false;
    if (nt <= 8 && et <= 1) *rc = launch_spec_graph_p<8, 1, 4>(a, bwd, stream);
    else if (nt <= 8) *rc = launch_spec_graph_p<8, 2, 4>(a, bwd, stream);
    else if (et <= 1) *rc = launch_spec_graph_p<12, 1, 4>(a, bwd, stream);
    else *rc = launch_spec_graph_p<12, 2, 4>(a, bwd, stream);
    return true;
  }
  if (nt <= 3 && et <= 1) *rc = launch_spec_graph<3, 1>(a, bwd, stream);
  else if (et <= 1) *rc = launch_spec_graph<4, 1>(a, bwd, stream);
  else if (nt <= 3) *rc = launch_spec_graph<3, 2>(a, bwd, stream);
  else *rc = launch_spec_graph<4, 2>(a, bwd, stream);
  return true;
}

bool try_spec_dense(const FilterArgs& a, int dh, bool bwd, hipStream_t stream, int* rc) {
  if (try_spec_graph(a, dh, bwd, stream, rc)) return true;
  if (dh == 16) return try_spec_dense_dh<16>(a, bwd, stream, rc);
  if (dh == 8) return try_spec_dense_dh<8>(a, bwd, stream, rc);
  if (dh == 4) return try_spec_dense_dh<4>(a, bwd, stream, rc);
  return false;
}

// ---- launchers ---------------------------------------------------------------------------

int check_filter(const FilterArgs& a, int dh, const void* p1, const void* p2) {
  FETA_REQUIRE(a.B > 0 && a.N > 0 && a.H > 0, "filter: empty shape");
  FETA_REQUIRE(a.N <= FETA_MAX_NODES, "filter: N=%d exceeds FETA_MAX_NODES", a.N);
  FETA_REQUIRE(dh == 4 || dh == 8 || dh == 16 || dh == 32 || dh == 64,
               "filter: head dim %d not in {4,8,16,32,64}", dh);
  FETA_REQUIRE(a.P >= 1 && a.P <= kMaxOrder, "filter: order P=%d not in [1,%d]", a.P, kMaxOrder);
  FETA_REQUIRE((a.xsb % 4) == 0 && (a.xsn % 4) == 0 && (a.ysb % 4) == 0 && (a.ysn % 4) == 0,
               "filter: strides must be multiples of 4 elements");
  FETA_REQUIRE(aligned16(p1) && aligned16(p2) && aligned16(a.coeff),
               "filter: token tensors and coeff must be 16-byte aligned");
  return FETA_OK;
}

template <template <int, int> class Launch>
int dispatch(const FilterArgs& a, int dh, int tiles, hipStream_t stream) {
  const int ct = (dh + 15) / 16;
  FETA_REQUIRE(tiles * ct <= 16, "filter: %d row tiles x %d column tiles exceed the register budget "
               "(need tiles*ceil(dh/16) <= 16)", tiles, ct);
#define FETA_T(D)                                             \
  if (tiles <= 3) return Launch<D, 3>::run(a, stream);        \
  if (tiles <= 4) return Launch<D, 4>::run(a, stream);        \
  if (tiles <= 8) return Launch<D, 8>::run(a, stream);        \
  return Launch<D, 16>::run(a, stream);
  switch (dh) {
    case 4: FETA_T(4)
    case 8: FETA_T(8)
    case 16: FETA_T(16)
    case 32:
      if (tiles <= 3) return Launch<32, 3>::run(a, stream);
      if (tiles <= 4) return Launch<32, 4>::run(a, stream);
      return Launch<32, 8>::run(a, stream);
    default:
      if (tiles <= 3) return Launch<64, 3>::run(a, stream);
      return Launch<64, 4>::run(a, stream);
  }
#undef FETA_T
}

#define FETA_LAUNCHER(NAME, KERNEL)                                            \
  template <int DH, int T>                                                     \
  struct NAME {                                                                \
    static int run(const FilterArgs& a, hipStream_t stream) {                  \
      const dim3 grid((a.total + kFWaves - 1) / kFWaves), block(64 * kFWaves); \
      auto kern = KERNEL<DH, T>;                                               \
      hipLaunchKernelGGL(kern, grid, block, 0, stream, a);                     \
      return check_launch(#KERNEL);                                            \
    }                                                                          \
  };
FETA_LAUNCHER(ChebFwd, cheb_fwd_kernel)
FETA_LAUNCHER(ChebBwd, cheb_bwd_kernel)
FETA_LAUNCHER(SpecFwd, spec_fwd_kernel)
FETA_LAUNCHER(SpecBwd, spec_bwd_kernel)

}  // namespace feta

using namespace feta;

extern "C" int feta_cheb_filter_fwd(const float* x, int64_t x_sb, int64_t x_sn, const float* lhat,
                                    const float* coeff, const float* bias, const int32_t* n_real,
                                    float* y, int64_t y_sb, int64_t y_sn, int B, int N, int H, int dh,
                                    int P, int heads_share_graph, feta_stream_t stream) {
  FilterArgs a{};
  a.x = x; a.lhat = lhat; a.coeff = coeff; a.bias = bias; a.n_real = n_real; a.y = y;
  a.xsb = x_sb; a.xsn = x_sn; a.ysb = y_sb; a.ysn = y_sn;
  a.B = B; a.N = N; a.H = H; a.P = P; a.share = heads_share_graph; a.total = B * H;
  FETA_REQUIRE(x && lhat && coeff && n_real && y, "cheb_filter_fwd: null pointer");
  int rc = check_filter(a, dh, x, y);
  if (rc != FETA_OK) return rc;
  return dispatch<ChebFwd>(a, dh, (N + 15) / 16, (hipStream_t)stream);
}

extern "C" int feta_cheb_filter_bwd(const float* x, int64_t x_sb, int64_t x_sn, const float* lhat,
                                    const float* coeff, const int32_t* n_real, const float* dy,
                                    int64_t y_sb, int64_t y_sn, float* dx, float* dcoeff,
                                    float* dbias_part, int B, int N, int H, int dh, int P,
                                    int heads_share_graph, feta_stream_t stream) {
  FilterArgs a{};
  a.x = x; a.lhat = lhat; a.coeff = coeff; a.n_real = n_real; a.dy = dy;
  a.dx = dx; a.dcoeff = dcoeff; a.dbias_part = dbias_part;
  a.xsb = x_sb; a.xsn = x_sn; a.ysb = y_sb; a.ysn = y_sn;
  a.B = B; a.N = N; a.H = H; a.P = P; a.share = heads_share_graph; a.total = B * H;
  FETA_REQUIRE(x && lhat && coeff && n_real && dy && dx && dcoeff && dbias_part,
               "cheb_filter_bwd: null pointer");
  int rc = check_filter(a, dh, x, dy);
  if (rc != FETA_OK) return rc;
  FETA_REQUIRE(aligned16(dx) && aligned16(dcoeff), "cheb_filter_bwd: outputs must be 16-byte aligned");
  return dispatch<ChebBwd>(a, dh, (N + 15) / 16, (hipStream_t)stream);
}

extern "C" int feta_spec_filter_fwd(const float* x, int64_t x_sb, int64_t x_sn, const float* u,
                                    const float* lam, const float* coeff, const float* bias,
                                    const int32_t* n_real, float* y, int64_t y_sb, int64_t y_sn,
                                    int B, int N, int H, int dh, int P, int K,
                                    int heads_share_graph, feta_stream_t stream) {
  FilterArgs a{};
  a.x = x; a.u = u; a.lam = lam; a.coeff = coeff; a.bias = bias; a.n_real = n_real; a.y = y;
  a.xsb = x_sb; a.xsn = x_sn; a.ysb = y_sb; a.ysn = y_sn;
  a.B = B; a.N = N; a.H = H; a.P = P; a.K = K; a.share = heads_share_graph; a.total = B * H;
  FETA_REQUIRE(x && u && lam && coeff && n_real && y, "spec_filter_fwd: null pointer");
  FETA_REQUIRE(K >= 1 && K <= FETA_MAX_NODES, "spec_filter_fwd: K=%d out of range", K);
  int rc = check_filter(a, dh, x, y);
  if (rc != FETA_OK) return rc;
  int rcd = FETA_OK;
  if (try_spec_dense(a, dh, false, (hipStream_t)stream, &rcd)) return rcd;
  return dispatch<SpecFwd>(a, dh, (K + 15) / 16, (hipStream_t)stream);
}

extern "C" int feta_spec_filter_bwd(const float* x, int64_t x_sb, int64_t x_sn, const float* u,
                                    const float* lam, const float* coeff, const int32_t* n_real,
                                    const float* dy, int64_t y_sb, int64_t y_sn, float* dx,
                                    float* dcoeff, float* dbias_part, int B, int N, int H, int dh,
                                    int P, int K, int heads_share_graph, feta_stream_t stream) {
  FilterArgs a{};
  a.x = x; a.u = u; a.lam = lam; a.coeff = coeff; a.n_real = n_real; a.dy = dy;
  a.dx = dx; a.dcoeff = dcoeff; a.dbias_part = dbias_part;
  a.xsb = x_sb; a.xsn = x_sn; a.ysb = y_sb; a.ysn = y_sn;
  a.B = B; a.N = N; a.H = H; a.P = P; a.K = K; a.share = heads_share_graph; a.total = B * H;
  FETA_REQUIRE(x && u && lam && coeff && n_real && dy && dx && dcoeff && dbias_part,
               "spec_filter_bwd: null pointer");
  FETA_REQUIRE(K >= 1 && K <= FETA_MAX_NODES, "spec_filter_bwd: K=%d out of range", K);
  int rc = check_filter(a, dh, x, dy);
  if (rc != FETA_OK) return rc;
  FETA_REQUIRE(aligned16(dx) && aligned16(dcoeff), "spec_filter_bwd: outputs must be 16-byte aligned");
  int rcd = FETA_OK;
  if (try_spec_dense(a, dh, true, (hipStream_t)stream, &rcd)) return rcd;
  return dispatch<SpecBwd>(a, dh, (K + 15) / 16, (hipStream_t)stream);
}

extern "C" int feta_spec_cat_supported(int N, int H, int dh, int P, int K, int heads_share_graph) {
  const int nt = (N + 15) / 16, et = (K + 15) / 16;
  return (heads_share_graph && H == 4 && dh == 16 && P == 4 && nt >= 1 && nt <= 8 && et <= 2 && (K & 3) == 0 && K >= 4) ? 1 : 0;
}

extern "C" int feta_spec_cat_bwd_supported(int N, int H, int dh, int P, int K, int heads_share_graph) {
  const int nt = (N + 15) / 16, et = (K + 15) / 16;
  return (heads_share_graph && H == 4 && dh == 16 && P == 4 && nt >= 1 && nt <= 4 && et <= 2 && (K & 3) == 0 && K >= 4) ? 1 : 0;
}

extern "C" int feta_spec_cat_bwd_rows(int B) { return B < 1 ? 0 : spec_cat_bwd_rows(B); }

extern "C" int feta_spec_filter_cat_bwd(const float* x, int64_t x_sb, int64_t x_sn, const float* u, const float* lam,
                                        const float* coeff, const int32_t* n_real, int64_t y_sb, int64_t y_sn, float* dx,
                                        float* dcoeff, float* dbias_part, int B, int N, int H, int dh, int P, int K,
                                        int heads_share_graph, const feta_spec_cat_grad* cat, feta_stream_t stream) {
  FilterArgs a{};
  a.x = x; a.u = u; a.lam = lam; a.coeff = coeff; a.n_real = n_real; a.dx = dx; a.dcoeff = dcoeff; a.dbias_part = dbias_part;
  a.xsb = x_sb; a.xsn = x_sn; a.ysb = y_sb; a.ysn = y_sn;
  a.B = B; a.N = N; a.H = H; a.P = P; a.K = K; a.share = heads_share_graph; a.total = B * H;
  FETA_REQUIRE(x && u && lam && coeff && n_real && dx && dcoeff && dbias_part && cat, "spec_filter_cat_bwd: null pointer");
  FETA_REQUIRE(feta_spec_cat_bwd_supported(N, H, dh, P, K, heads_share_graph),
               "spec_filter_cat_bwd: needs 4 heads x 16, order 4, N <= 64, K <= 32 (multiple of 4), every head on the graph");
  int rc = check_filter(a, dh, x, dx);
  if (rc != FETA_OK) return rc;
  FETA_REQUIRE(cat->dout && cat->y2 && cat->filt && cat->w_cat && cat->dxn && cat->partial,
               "spec_filter_cat_bwd: dout, y2, filt, w_cat, dxn, partial");
  FETA_REQUIRE(cat->partial_ld >= 64 * 128 + 64, "spec_filter_cat_bwd: partial_ld %lld < 64 * 128 + 64", (long long)cat->partial_ld);
  FETA_REQUIRE(!cat->gs || cat->y2_bn, "spec_filter_cat_bwd: gs (BatchNorm-backward sums) needs the published block y2_bn");
  FETA_REQUIRE(aligned16(u) && aligned16(cat->dout) && aligned16(cat->dxn) && aligned16(cat->filt) && (cat->y2_sb % 4) == 0 &&
                   (cat->y2_sn % 4) == 0 && (y_sb % 4) == 0 && (y_sn % 4) == 0,
               "spec_filter_cat_bwd: 16-byte aligned tensors, strides multiples of 4 elements");
  CatGradArgs c{};
  c.dout = cat->dout; c.y2 = cat->y2; c.y2sb = cat->y2_sb; c.y2sn = cat->y2_sn; c.y2_bn = cat->y2_bn; c.filt = cat->filt;
  c.w_cat = cat->w_cat; c.dxn = cat->dxn; c.gs = cat->gs; c.partial = cat->partial; c.partial_ld = cat->partial_ld;
  const int nt = (N + 15) / 16, et = (K + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  if (nt <= 3) return et <= 1 ? launch_spec_cat_bwd<3, 1>(a, c, st) : launch_spec_cat_bwd<3, 2>(a, c, st);
  return et <= 1 ? launch_spec_cat_bwd<4, 1>(a, c, st) : launch_spec_cat_bwd<4, 2>(a, c, st);
}

extern "C" int feta_spec_filter_cat_fwd(const float* x, int64_t x_sb, int64_t x_sn, const float* u, const float* lam,
                                        const float* coeff, const float* bias, const int32_t* n_real, float* y,
                                        int64_t y_sb, int64_t y_sn, int B, int N, int H, int dh, int P, int K,
                                        int heads_share_graph, const feta_spec_cat* cat, feta_stream_t stream) {
  FilterArgs a{};
  a.x = x; a.u = u; a.lam = lam; a.coeff = coeff; a.bias = bias; a.n_real = n_real; a.y = y;
  a.xsb = x_sb; a.xsn = x_sn; a.ysb = y_sb; a.ysn = y_sn;
  a.B = B; a.N = N; a.H = H; a.P = P; a.K = K; a.share = heads_share_graph; a.total = B * H;
  FETA_REQUIRE(x && u && lam && coeff && n_real && y && cat, "spec_filter_cat_fwd: null pointer");
  FETA_REQUIRE(feta_spec_cat_supported(N, H, dh, P, K, heads_share_graph),
               "spec_filter_cat_fwd: needs 4 heads x 16, order 4, N <= 128, K <= 32 (multiple of 4), every head on the graph");
  int rc = check_filter(a, dh, x, y);
  if (rc != FETA_OK) return rc;
  FETA_REQUIRE(cat->y2 && cat->w_cat && cat->out, "spec_filter_cat_fwd: y2, w_cat, out");
  FETA_REQUIRE(!(cat->y2_bn && cat->y2_stats), "spec_filter_cat_fwd: y2_bn and y2_stats exclude each other");
  FETA_REQUIRE(!cat->y2_stats || (cat->gamma && cat->beta && cat->bn_out && cat->Gx > 0 && cat->M > 0),
               "spec_filter_cat_fwd: y2_stats needs gamma, beta, bn_out, Gx, M");
  FETA_REQUIRE(aligned16(u) && aligned16(cat->y2) && aligned16(cat->w_cat) && aligned16(cat->out) && aligned16(cat->y2_stats) &&
                   aligned16(bias) && (cat->y2_sb % 4) == 0 && (cat->y2_sn % 4) == 0,
               "spec_filter_cat_fwd: 16-byte aligned tensors, strides multiples of 4 elements");
  CatArgs c{};
  c.y2 = cat->y2; c.y2sb = cat->y2_sb; c.y2sn = cat->y2_sn; c.y2_bn = cat->y2_bn; c.y2_stats = cat->y2_stats; c.Gx = cat->Gx;
  c.gamma = cat->gamma; c.beta = cat->beta; c.bn_out = cat->bn_out; c.rmean = cat->rmean; c.rvar = cat->rvar;
  c.nbt = cat->nbt; c.momentum = cat->momentum; c.eps = cat->eps; c.M = cat->M;
  c.w_cat = cat->w_cat; c.b_cat = cat->b_cat; c.out = cat->out;
  const int nt = (N + 15) / 16, et = (K + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  if (nt <= 3) return et <= 1 ? launch_spec_cat_fwd<3, 1>(a, c, st) : launch_spec_cat_fwd<3, 2>(a, c, st);
  if (nt <= 4) return et <= 1 ? launch_spec_cat_fwd<4, 1>(a, c, st) : launch_spec_cat_fwd<4, 2>(a, c, st);
  return et <= 1 ? launch_spec_cat_fwd<8, 1>(a, c, st) : launch_spec_cat_fwd<8, 2>(a, c, st);
}
