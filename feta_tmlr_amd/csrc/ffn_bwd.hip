// Backward of the feed-forward half of DiffTransformerEncoderLayer as ONE launch (d_model = 64,
// dim_feedforward in {64, 128}): the two feta_rowlin_bwd_ex launches of linear2 and linear1
// (contract transformer/models.py:166-167; body per upstream GraphiT, README.md:129:
// y2 = x1 + linear2(relu(linear1(x1))), norm2) become one kernel whose workgroups take roles:
//
//   g2  = BN2-backward(dy)            (or dy itself: LayerNorm stack)      [M, 64]
//   dh  = (g2 W2) * [h > 0]                                                 [M, FF]
//   dx1 = g2 + dh W1                  (+ partial sums of BN1's backward)    [M, 64]
//   dW2 = g2^T h,  db2 = colsum g2,   dW1 = dh^T x1,  db1 = colsum dh       (split-K partials per row chunk)
//
//   X role  one workgroup per 32-row block: g2 tile staged once in LDS (all transforms applied), dh tile computed
//           into LDS (never written to HBM - the two-launch form wrote and re-read M x FF floats), dx1 from it;
//           a wave owns one 16-column slice of W1 / FF/64 slices of W2 in registers (as feta_rowlin_bwd's dX role).
//   W role  one workgroup per (64-row chunk, 32-column slice of the hidden units): recomputes ITS slice of dh from
//           the staged g2 tile (128 MFMAs: cheaper than a hand-over between workgroups, which would need a second
//           launch or a grid-wide wait), then both weight-gradient products from LDS tiles; one partial row per
//           chunk in the caller's [chunks, ld] buffer, columns [dW2 | db2 | dW1 | db1] (the slots the fused stack
//           reduces with one feta_colsum).
// Every launch of a captured step costs ~4.5 us whatever its size (profiles/r02_*): this removes one per layer.
#include <cstdlib>

#include "feta_abi_common.h"
#include "feta_coeff.h"
#include "feta_ln.h"
#include "feta_lp.h"
#include "feta_rowops.h"

namespace feta {

typedef feta_ffn_grad FfnGradArgs;  // include/feta_hip.h

constexpr int kFbD = 64;
constexpr int kFbRows = 32;   // rows of an X-role block
constexpr int kFbSlice = 32;  // hidden units of a W-role workgroup

struct FfnBwdGeom {
  int XB;   // X-role workgroups (= partial rows of sum_out); they walk the 32-row blocks
  int RC;   // row chunks of the weight gradient (feta_rowlin_chunks)
  int per;  // 16-row blocks per chunk
  int NS;   // hidden-unit slices per chunk
  int xcd;  // 1: XCD-aware order of the workgroups (see ffn_bwd_kernel)
  int wxcd; // 1: XCD-aware order of the W-role workgroups of a capped grid
  int cxw;  // > 0: capped grid with the X role walking chunk-wise, cxw X walkers per chunk (wxcd is then not used)
  int main_grid;   // workgroups of the two roles above
  int role, role_lead;   // the coefficient generator's backward (feta_coeff.h): its workgroups, and how many grid slots lead the main ones (0: they trail)
};

inline int ffn_bwd_xblocks(int M) {
  int cap = 512;
  if (const char* e = getenv("FETA_FFN_MAX_GRID")) cap = atoi(e) > 0 ? atoi(e) : cap;
  const int nblk = (M + kFbRows - 1) / kFbRows;
  return nblk < cap ? nblk : cap;
}

// g2(row, o .. o+NV-1) from dy (and y2 when the BatchNorm backward is folded in), in place in v;
// gv = [5][64]: scale, mean, rstd, m1, m2
template <int NV>
__device__ __forceinline__ void g2_of(float (&v)[NV], const float (&yy)[NV], const float* gv, int o, bool gbn) {
  if (gbn) {
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      const float xh = (yy[s] - gv[kFbD + o + s]) * gv[2 * kFbD + o + s];
      v[s] = gv[o + s] * (v[s] - gv[3 * kFbD + o + s] - xh * gv[4 * kFbD + o + s]);
    }
  }
}

// LDS row paddings of the W role's tiles (elements): their operands are read down columns as well as along rows
template <class T> struct FbPad { static constexpr int W = 16; };
template <> struct FbPad<bf16_t> { static constexpr int W = 8; };

template <class T>
inline size_t ffn_bwd_lds_bytes(int FF, bool gbn) {
  const size_t x_lds = kFbRows * (kFbD + Lp<T>::PAD) + kFbRows * (FF + Lp<T>::PAD);
  const size_t w_lds = 2 * 64 * (kFbD + FbPad<T>::W) + 2 * 64 * (kFbSlice + FbPad<T>::W);
  return sizeof(T) * (x_lds > w_lds ? x_lds : w_lds) + sizeof(float) * (gbn ? 5 * kFbD + reduce_scratch_floats(kFbD) : 0);
}

// T: storage type of dy, dy_b, g_y, h, x, dx and of the LDS tiles (feta_lp.h); weights, parameter blocks, partial
// sums and the weight-gradient partial rows: fp32.
template <class T, int FF>
__global__ __launch_bounds__(kRowThreads) void ffn_bwd_kernel(FfnGradArgs a, FfnBwdGeom ge, CoeffBwdRole cb) {
  typedef Lp<T> L;
  typedef typename L::Op Op;
  typedef typename L::Vec Vec;
  constexpr int D = kFbD, NJ2 = FF / 16, VEC = L::VEC, RV = D / VEC;
  // Extra workgroups: the backward of the coefficient generator (feta_coeff.h) - it depends on the filter stage only, and
  // this is the first launch of the layer stack's backward.  They LEAD the grid (ge.role_lead, a multiple of 8 so that the
  // XCD of every main workgroup stays what the orders below assume): the kernel's ~200 registers allow two workgroups per
  // CU, 512 at a time - behind the main roles the generator's 512 short workgroups were a second round that started when the
  // main ones (13 us) had finished; in front they are gone after ~3 us and the main workgroups move into their slots.
  int bid = (int)blockIdx.x;
  if (bid < ge.role_lead || bid >= ge.role_lead + ge.main_grid) {
    const int r = bid < ge.role_lead ? bid : bid - ge.main_grid;
    const int nbx = (cb.C + kCoeffThreads - 1) / kCoeffThreads;
    if (r < ge.role)
      coeff_bwd_body(cb.cj, cb.n_real, cb.s, cb.gbias, cb.dpooled, cb.partial, cb.B, cb.N, cb.H, cb.C, cb.G, r % nbx, r / nbx);
    return;
  }
  bid -= ge.role_lead;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lq = lane & 15, g = lane >> 4;
  const T* gdy = reinterpret_cast<const T*>(a.dy);
  const T* gdyb = reinterpret_cast<const T*>(a.dy_b);
  const T* ggy = reinterpret_cast<const T*>(a.g_y);
  const T* ghh = reinterpret_cast<const T*>(a.h);
  const T* gxx = reinterpret_cast<const T*>(a.x);
  T* gdx = reinterpret_cast<T*>(a.dx);
  // Which piece of work this workgroup is.  A 64-row chunk is touched by six workgroups (its two X-role blocks and the
  // four hidden-unit slices of the W role), and workgroups are dealt round-robin to the 8 XCDs, each with its own L2:
  // in launch order (X blocks, then W slices) the six sat on up to six XCDs and the chunk's rows of dy, y2, y1 left HBM
  // up to six times (2.3x the algorithmic bytes by the counters).  With ge.xcd the grid is read as
  // (chunk group, item, XCD): all six items of chunk 8 * group + XCD run on that XCD.
  int xblk, wbi;
  if (ge.xcd) {
    const int xcd = bid & 7, v = bid >> 3, items = 2 + ge.NS;
    const int item = v % items, rc = (v / items) * 8 + xcd;
    if (rc >= ge.RC) return;
    xblk = item < 2 ? 2 * rc + item : -1;
    wbi = item < 2 ? -1 : rc * ge.NS + (item - 2);
    if (item < 2 && xblk >= ge.XB) return;
  } else {
    xblk = bid < ge.XB ? bid : -1;
    wbi = bid - ge.XB;
    if (wbi >= 0 && ge.wxcd) {
      // capped grids (large batches): the NS hidden-unit slices of a row chunk read the same rows of dy, y2, x - in
      // launch order they sat on NS different XCDs and every one fetched the chunk from HBM (3.3x the algorithmic
      // bytes at B = 16384, profiles/r03_traffic_b16384.json).  Read as (chunk group, slice, XCD): the slices of chunk
      // 8 * group + XCD run on that XCD, 8 workgroup ids apart, and meet in its L2 (1.9x).
      const int xcd = wbi & 7, v = wbi >> 3;
      const int si_ = v % ge.NS, rc_ = (v / ge.NS) * 8 + xcd;
      if (rc_ >= ge.RC) return;
      wbi = rc_ * ge.NS + si_;
    }
  }
  // ge.cxw > 0 (capped grid, whole chunks per XCD): the X role walks ITS chunk too.  The grid is read as (chunk group,
  // item, XCD) with cxw X walkers and NS W slices per chunk; walker i of chunk rc takes the chunk's 32-row blocks
  // i, i + cxw, ... - at the pace of the W slices, which pass over the same rows 64 at a time on the same XCD.
  int x_lo = xblk, x_hi = (a.M + kFbRows - 1) / kFbRows, x_step = ge.XB;
  if (ge.cxw > 0) {
    const int xcd = bid & 7, v = bid >> 3, items = ge.cxw + ge.NS;
    const int item = v % items, rc = (v / items) * 8 + xcd;
    if (rc >= ge.RC) return;
    if (item < ge.cxw) {
      xblk = rc * ge.cxw + item;                 // (its row of sum_out)
      wbi = -1;
      x_lo = rc * (ge.per / 2) + item;
      x_hi = min((rc + 1) * (ge.per / 2), x_hi);
      x_step = ge.cxw;
    } else {
      xblk = -1;
      wbi = rc * ge.NS + (item - ge.cxw);
    }
  }
  float* gv = feta_lds;   // [5][64]
  float* after = gv;
  const bool gbn = a.g_bn != nullptr;
  // LayerNorm stack (feta_ln.h): dy is the gradient w.r.t. LN2(y2) - its LayerNorm backward is taken per row where the
  // gradient rows are staged (both roles), x = LN1(y1) per row where the x rows are staged (W role).  A thread of a
  // staging loop holds the same VEC columns of every row it touches: its gamma / beta slices live in registers.
  const bool gln = a.g_ln_gamma != nullptr, xln = a.x_ln_gamma != nullptr;
  const float ln_eps = a.ln_eps;
  float gl2[L::VEC], gl1[L::VEC], bl1[L::VEC];
  {
    const int c0v = L::VEC * (tid % (kFbD / L::VEC));
#pragma unroll
    for (int e = 0; e < L::VEC; ++e) {
      gl2[e] = gln ? a.g_ln_gamma[c0v + e] : 1.0f;
      gl1[e] = xln ? a.x_ln_gamma[c0v + e] : 1.0f;
      bl1[e] = xln ? a.x_ln_beta[c0v + e] : 0.0f;
    }
  }
  if (gbn) {
    float* scr = gv + 5 * D;
    after = scr;
    const int cpre = min(tid, D - 1);
    const float bn_scale = a.g_bn[cpre], bn_mean = a.g_bn[2 * D + cpre], bn_rstd = a.g_bn[3 * D + cpre];
    if (a.g_sum != nullptr) {
      reduce_partials_t<kRowThreads, 32>(a.g_sum, a.Gs, D, scr + 2 * D, scr);   // (256 rows in ONE batch: feta_rowops.h)
      for (int c = tid; c < D; c += kRowThreads) {
        gv[3 * D + c] = scr[c] / (float)a.M;
        gv[4 * D + c] = scr[D + c] / (float)a.M;
        if (bid == 0) {
          if (a.dbeta != nullptr) a.dbeta[c] = scr[c];
          if (a.dgamma != nullptr) a.dgamma[c] = scr[D + c];
          if (a.g_fin_out != nullptr) {
            a.g_fin_out[c] = gv[3 * D + c];
            a.g_fin_out[D + c] = gv[4 * D + c];
          }
        }
      }
    } else {
      for (int c = tid; c < 2 * D; c += kRowThreads) gv[3 * D + c] = a.g_fin[c];
    }
    if (tid < D) {
      gv[tid] = bn_scale;
      gv[D + tid] = bn_mean;
      gv[2 * D + tid] = bn_rstd;
    }
    __syncthreads();
  }
  // gradient vector `idx` of a row tile: dy (+ dy_b) and y2, as fp32
  // (captures local pointers only: a lambda that captures the argument struct `a` by reference here made the compiler
  // keep a copy of the whole struct in private memory)
  const bool two_parts = a.dy_b != nullptr;
  const bool has_gy = a.g_y != nullptr;
  const bool g_f32 = a.g_f32 != 0 && sizeof(T) != sizeof(float);   // dy, g_y fp32 behind a bf16 stack (its last layer)
  const float* fdy = a.dy;
  const float* fgy = a.g_y;
  auto load_g = [gdy, gdyb, ggy, has_gy, two_parts, g_f32, fdy, fgy](int64_t off, float (&dv)[VEC], float (&yv)[VEC]) {
    const bool gbn = has_gy;   // (the pre-norm rows are read for the BatchNorm and for the LayerNorm backward alike)
    if (g_f32) {
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        const float4 d4 = *reinterpret_cast<const float4*>(fdy + off + e);
        const float4 y4 = *reinterpret_cast<const float4*>((gbn ? fgy : fdy) + off + e);
        dv[e] = d4.x; dv[e + 1] = d4.y; dv[e + 2] = d4.z; dv[e + 3] = d4.w;
        yv[e] = y4.x; yv[e + 1] = y4.y; yv[e + 2] = y4.z; yv[e + 3] = y4.w;
      }
      return;
    }
    L::unpack(L::ldv(gdy + off), dv);
    if (two_parts) {   // the gradient arrives in two parts (feta_attn_block_bwd, SPLIT form)
      float d2[VEC];
      L::unpack(L::ldv(gdyb + off), d2);
#pragma unroll
      for (int e = 0; e < VEC; ++e) dv[e] += d2[e];
    }
    L::unpack(L::ldv((gbn ? ggy : gdy) + off), yv);
  };

  if (xblk >= 0) {
    // ================================ X role: dh tile -> dx1 ================================================
    constexpr int GP = D + L::PAD, DP = FF + L::PAD, CT2 = FF / 64;   // CT2 column tiles of dh per wave
    T* gt = reinterpret_cast<T*>(after);   // [32][GP]  g2
    T* dht = gt + kFbRows * GP;            // [32][DP]  dh
    // weight slices of this wave, for the whole launch
    Op wA2[CT2][4];                 // W2[o = 16j+4g+s][c = 16 (w CT2 + t) + lq]
#pragma unroll
    for (int t = 0; t < CT2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* p = a.w2 + (int64_t)(16 * j + 4 * g) * FF + 16 * (w * CT2 + t) + lq;
        wA2[t][j] = L::mk(p[0], p[FF], p[2 * FF], p[3 * FF]);
      }
    Op wA1[NJ2];                    // W1[c = 16j+4g+s][k = 16w + lq]
#pragma unroll
    for (int j = 0; j < NJ2; ++j) {
      const float* p = a.w1 + (int64_t)(16 * j + 4 * g) * D + 16 * w + lq;
      wA1[j] = L::mk(p[0], p[D], p[2 * D], p[3 * D]);
    }
    const bool want_sums = a.sum_out != nullptr;
    float mean1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, rstd1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (want_sums) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        mean1[r] = a.x_bn[2 * D + 16 * w + 4 * g + r];
        rstd1[r] = a.x_bn[3 * D + 16 * w + 4 * g + r];
      }
    }
    float sum1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sum2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int nblk = (a.M + kFbRows - 1) / kFbRows;
    const int row_last = a.M - 1;
    constexpr int NU = kFbRows * RV / kRowThreads;   // vectors of the gradient tile per thread
    const int lane0 = lane;
    for (int blk = x_lo; blk < x_hi; blk += x_step) {
      // (the lane id is laundered once per row block: csrc/block_bwd.hip)
      int lane_l = lane0;
      FETA_OPAQUE_LANE(lane_l);
      const int lane = lane_l, tid = (w << 6) | lane, lq = lane & 15, g = lane >> 4;
      const int r0 = blk * kFbRows;
      if (blk != x_lo) __syncthreads();   // the tiles of the previous block have been consumed
      // requests of the block: gradient tile, relu operands, residual rows of the sums
      float dv[NU][VEC], yv[NU][VEC];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int idx = tid + u * kRowThreads, rr = idx / RV, c4 = idx % RV;
        load_g((int64_t)min(r0 + rr, row_last) * D + VEC * c4, dv[u], yv[u]);
      }
      float hv[2][CT2][4], sy[2][4];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int64_t rowc = min(r0 + 16 * rt + lq, row_last);
#pragma unroll
        for (int t = 0; t < CT2; ++t) L::ld4(ghh + rowc * FF + 16 * (w * CT2 + t) + 4 * g, hv[rt][t]);
        L::ld4(gxx + rowc * D + 16 * w + 4 * g, sy[rt]);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int idx = tid + u * kRowThreads, rr = idx / RV, c4 = idx % RV;
        g2_of<VEC>(dv[u], yv[u], gv, VEC * c4, gbn);
        const bool ok = r0 + rr < a.M;
        if (gln) {
          float dg_[VEC], db_[VEC];   // (the W role emits dgamma2 / dbeta2)
#pragma unroll
          for (int e = 0; e < VEC; ++e) dg_[e] = db_[e] = 0.0f;
          ln_backward<VEC>(dv[u], yv[u], gl2, ln_eps, ok, dg_, db_);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) dv[u][e] = ok ? dv[u][e] : 0.0f;
        L::stv(gt + rr * GP + VEC * c4, L::pack(dv[u]));
      }
      __syncthreads();
      // dh^T tiles (c = 16 ct + 4g + r, row = 16 rt + lq) = sum_o W2[o][c] g2[row][o], masked by h > 0
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        RowOp<T, D> gf;
        load_row_op<T, D>(gf, gt + (16 * rt + lq) * GP, g);
#pragma unroll
        for (int t = 0; t < CT2; ++t) {
          f32x4 acc = zero4();
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = L::mma(wA2[t][j], gf.o[j], acc);
          L::st4(dht + (16 * rt + lq) * DP + 16 * (w * CT2 + t) + 4 * g,
                 hv[rt][t][0] > 0.0f ? acc[0] : 0.0f, hv[rt][t][1] > 0.0f ? acc[1] : 0.0f,
                 hv[rt][t][2] > 0.0f ? acc[2] : 0.0f, hv[rt][t][3] > 0.0f ? acc[3] : 0.0f);
        }
      }
      __syncthreads();
      // dx1^T tiles (k = 16w + 4g + r, row) = sum_c W1[c][k] dh[row][c]  + g2[row][k]
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        RowOp<T, FF> df;
        load_row_op<T, FF>(df, dht + (16 * rt + lq) * DP, g);
        f32x4 acc = zero4();
#pragma unroll
        for (int j = 0; j < NJ2; ++j) acc = L::mma(wA1[j], df.o[j], acc);
        float res[4];
        L::ld4(gt + (16 * rt + lq) * GP + 16 * w + 4 * g, res);
        const float v[4] = {acc[0] + res[0], acc[1] + res[1], acc[2] + res[2], acc[3] + res[3]};
        const int row = r0 + 16 * rt + lq;
        const bool rok = row < a.M;
        if (rok) L::st4(gdx + (int64_t)row * D + 16 * w + 4 * g, v[0], v[1], v[2], v[3]);
        if (want_sums) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xh = (sy[rt][r] - mean1[r]) * rstd1[r];
            const float s1 = rok ? v[r] : 0.0f;
            sum1[r] += s1;
            sum2[r] += s1 * xh;
          }
        }
      }
    }
    if (want_sums) {   // a wave's 16 columns are its own: no cross-wave reduction
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = row16_sum(sum1[r]), s2 = row16_sum(sum2[r]);
        if (lq == 0) {
          a.sum_out[((int64_t)xblk * 2 + 0) * D + 16 * w + 4 * g + r] = s1;
          a.sum_out[((int64_t)xblk * 2 + 1) * D + 16 * w + 4 * g + r] = s2;
        }
      }
    }
    return;
  }

  // ================================== W role: dW2 | db2 | dW1 | db1 of one (chunk, hidden slice) ===================
  constexpr int GPW = D + FbPad<T>::W, HP = kFbSlice + FbPad<T>::W, HV = kFbSlice / VEC;
  const int bi = wbi;
  const int si = bi % ge.NS, rc = bi / ge.NS;
  const int c0 = kFbSlice * si;
  T* gt = reinterpret_cast<T*>(after);   // [64][GPW] g2
  T* xt = gt + 64 * GPW;     // [64][GPW] x1 (seen through its BatchNorm)
  T* hs = xt + 64 * GPW;     // [64][HP]  h slice
  T* dhs = hs + 64 * HP;     // [64][HP]  dh slice
  Op wA2[2][4];              // W2[o = 16j+4g+s][c = c0 + 16t + lq]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* p = a.w2 + (int64_t)(16 * j + 4 * g) * FF + c0 + 16 * t + lq;
      wA2[t][j] = L::mk(p[0], p[FF], p[2 * FF], p[3 * FF]);
    }
  const int row_lo = rc * ge.per * 16, row_hi = min((rc + 1) * ge.per * 16, a.M);
  const int row_last = max(row_hi - 1, 0);
  f32x4 aW2[2] = {zero4(), zero4()}, aW1[2] = {zero4(), zero4()};
  // bias gradients = column sums of g2 / dh: taken from the fp32 values BEFORE they are rounded into the tiles (behind a
  // BatchNorm backward the true column sum of g2 is zero: the rounding errors of a bf16 tile would be all that is left)
  float cs2[VEC], cs1[2][4];
#pragma unroll
  for (int e = 0; e < VEC; ++e) cs2[e] = 0.0f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs1[t][r] = 0.0f;
  const bool xbn = a.x_bn != nullptr;
  float dgam[VEC], dbet[VEC];   // LayerNorm stack: sums of (dy xhat2, dy) over the chunk's rows, this thread's columns
#pragma unroll
  for (int e = 0; e < VEC; ++e) dgam[e] = dbet[e] = 0.0f;
  constexpr int NG = 64 * RV / kRowThreads, NH = 64 * HV / kRowThreads;   // vectors per thread: g2 / x1 tiles, h slice
  for (int r0 = row_lo; r0 < row_hi; r0 += 64) {
    if (r0 > row_lo) __syncthreads();
    // stage g2 [64 x 64], x1 [64 x 64] and the h slice [64 x 32]: every load of the pass is issued before the first
    // LDS store
    float dv[NG][VEC], yv[NG][VEC];
    Vec xv[NG], hq[NH];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int idx = tid + u * kRowThreads, rr = idx / RV, c4 = idx % RV;
      const int64_t off = (int64_t)min(r0 + rr, row_last) * D + VEC * c4;
      load_g(off, dv[u], yv[u]);
      xv[u] = L::ldv(gxx + off);
    }
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int idx = tid + u * kRowThreads, rr = idx / HV, c4 = idx % HV;
      hq[u] = L::ldv(ghh + (int64_t)min(r0 + rr, row_last) * FF + c0 + VEC * c4);
    }
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int idx = tid + u * kRowThreads, rr = idx / RV, c4 = idx % RV;
      const bool ok = r0 + rr < row_hi;
      g2_of<VEC>(dv[u], yv[u], gv, VEC * c4, gbn);
      if (gln) ln_backward<VEC>(dv[u], yv[u], gl2, ln_eps, ok, dgam, dbet);
      float x[VEC];
      L::unpack(xv[u], x);
      if (xln) ln_apply<VEC>(x, gl1, bl1, ln_eps);
      if (xbn) {
#pragma unroll
        for (int e4 = 0; e4 < VEC; e4 += 4) {
          const float4 sc = *reinterpret_cast<const float4*>(a.x_bn + VEC * c4 + e4);
          const float4 sh = *reinterpret_cast<const float4*>(a.x_bn + D + VEC * c4 + e4);
          x[e4] = x[e4] * sc.x + sh.x;
          x[e4 + 1] = x[e4 + 1] * sc.y + sh.y;
          x[e4 + 2] = x[e4 + 2] * sc.z + sh.z;
          x[e4 + 3] = x[e4 + 3] * sc.w + sh.w;
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        dv[u][e] = ok ? dv[u][e] : 0.0f;
        cs2[e] += dv[u][e];
        x[e] = ok ? x[e] : 0.0f;
      }
      L::stv(gt + rr * GPW + VEC * c4, L::pack(dv[u]));
      L::stv(xt + rr * GPW + VEC * c4, L::pack(x));
    }
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int idx = tid + u * kRowThreads, rr = idx / HV, c4 = idx % HV;
      const bool ok = r0 + rr < row_hi;
      float hf[VEC];
      L::unpack(hq[u], hf);
#pragma unroll
      for (int e = 0; e < VEC; ++e) hf[e] = ok ? hf[e] : 0.0f;
      L::stv(hs + rr * HP + VEC * c4, L::pack(hf));
    }
    __syncthreads();
    // this wave's row tile of the dh slice: (c = c0 + 16t + 4g + r, row = 16w + lq)
    {
      RowOp<T, D> gf;
      load_row_op<T, D>(gf, gt + (16 * w + lq) * GPW, g);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = zero4();
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = L::mma(wA2[t][j], gf.o[j], acc);
        float hh[4];
        L::ld4(hs + (16 * w + lq) * HP + 16 * t + 4 * g, hh);
        const float dm[4] = {hh[0] > 0.0f ? acc[0] : 0.0f, hh[1] > 0.0f ? acc[1] : 0.0f, hh[2] > 0.0f ? acc[2] : 0.0f,
                             hh[3] > 0.0f ? acc[3] : 0.0f};
        L::st4(dhs + (16 * w + lq) * HP + 16 * t + 4 * g, dm[0], dm[1], dm[2], dm[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) cs1[t][r] += dm[r];
      }
    }
    __syncthreads();
    // dW2[o = 16w + 4g' + r][c = c0 + 16ct + lq] += g2[row][o] h[row][c];  dW1[c][k = 16w + lq] += dh[row][c] x1[row][k]
    // (contraction over the 64 rows of the pass: k-step s of group q is row 16 q + 4 s + g)
#pragma unroll 2
    for (int q = 0; q < 4; ++q) {
      const int rr = 16 * q + g;
      const Op ga = L::gather(gt + rr * GPW + 16 * w + lq, 4 * GPW);
      const Op xb = L::gather(xt + rr * GPW + 16 * w + lq, 4 * GPW);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const Op hb = L::gather(hs + rr * HP + 16 * ct + lq, 4 * HP);
        const Op da = L::gather(dhs + rr * HP + 16 * ct + lq, 4 * HP);
        aW2[ct] = L::mma(ga, hb, aW2[ct]);
        aW1[ct] = L::mma(da, xb, aW1[ct]);
      }
    }
  }
  float* p = a.partial + (int64_t)rc * (a.partial_ld > 0 ? (int64_t)a.partial_ld
                                                         : (int64_t)(2 * D * FF + D + FF + (gln ? 2 * D : 0)));
  float* pW2 = p;
  float* pb2 = p + D * FF;
  float* pW1 = pb2 + D;
  float* pb1 = pW1 + FF * D;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pW2[(int64_t)(16 * w + 4 * g + r) * FF + c0 + 16 * ct + lq] = aW2[ct][r];
      pW1[(int64_t)(c0 + 16 * ct + 4 * g + r) * D + 16 * w + lq] = aW1[ct][r];
    }
  // db2 | db1: the per-thread fp32 column sums meet in LDS (the tiles are no longer needed), fixed order
  __syncthreads();
  float* red2 = after;                        // [256][VEC]: thread tid holds columns VEC (tid % RV) ..
  float* red1 = after + kRowThreads * VEC;    // [4 waves][2 tiles][16]
  float* red3 = red1 + 4 * 2 * 16;            // [256][VEC] dgamma2, [256][VEC] dbeta2 (LayerNorm stack)
  float* red4 = red3 + kRowThreads * VEC;
  if (gln && si == 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      red3[tid * VEC + e] = dgam[e];
      red4[tid * VEC + e] = dbet[e];
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red2[tid * VEC + e] = cs2[e];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sv = row16_sum(cs1[t][r]);
      if (lq == 0) red1[(w * 2 + t) * 16 + 4 * g + r] = sv;
    }
  __syncthreads();
  if (si == 0 && tid < D) {
    const int c4 = tid / VEC, e = tid % VEC;
    float sv = 0.0f;
    for (int k = 0; k < kRowThreads / RV; ++k) sv += red2[(c4 + RV * k) * VEC + e];
    pb2[tid] = sv;
    if (gln) {   // [dgamma2 | dbeta2] behind db1
      float sg = 0.0f, sb = 0.0f;
      for (int k = 0; k < kRowThreads / RV; ++k) {
        sg += red3[(c4 + RV * k) * VEC + e];
        sb += red4[(c4 + RV * k) * VEC + e];
      }
      pb1[FF + tid] = sg;
      pb1[FF + D + tid] = sb;
    }
  }
  if (tid < kFbSlice) {
    const int t = tid >> 4, cc = tid & 15;
    pb1[c0 + tid] = (red1[(0 * 2 + t) * 16 + cc] + red1[(1 * 2 + t) * 16 + cc]) +
                    (red1[(2 * 2 + t) * 16 + cc] + red1[(3 * 2 + t) * 16 + cc]);
  }
}

extern int row_chunks(int M);   // rowwise.hip

// Workgroups of the two main roles for RC split-K chunks of M rows (the same arithmetic as launch_ffn_bwd below).
static int ffn_bwd_main_grid(int M, int RC, int NS) {
  const int XB = ffn_bwd_xblocks(M), nblk = (M + kFbRows - 1) / kFbRows, nrb16 = (M + 15) / 16;
  const int per = (nrb16 + RC - 1) / RC;
  const bool xcd = XB == nblk && per == 4;
  return xcd ? 8 * ((RC + 7) / 8) * (2 + NS) : XB + 8 * ((RC + 7) / 8) * NS;
}
// Split-K chunks (= partial rows) of a launch: the 64-row chunks of the row-wise kernels - unless that makes the grid
// just exceed one round of resident workgroups (the kernel's ~200 registers allow two per CU, 512) and chunks of twice the
// rows fit: config 4 (M = 8192) was 256 X blocks + 128 x 4 W slices = 768 workgroups, 21.7 us; with 64 chunks of 128 rows it
// is 512, 17.5 us (PATTERN B = 64, N_pad = 128: 0.405 -> 0.398 ms).  FETA_FFN_BWD_RC_HALF=0: off (A/B).
static int ffn_bwd_chunks_for(int M, int ff) {
  const int rc = row_chunks(M), ns = ff / kFbSlice;
  if (const char* e = getenv("FETA_FFN_BWD_RC_HALF"))
    if (atoi(e) == 0) return rc;
  const int half = (rc + 1) / 2;
  if (rc >= 2 && ffn_bwd_main_grid(M, rc, ns) > 512 && ffn_bwd_main_grid(M, half, ns) <= 512) return half;
  return rc;
}

template <class T, int FF>
int launch_ffn_bwd(const FfnGradArgs& a, const CoeffBwdRole& cb, hipStream_t stream) {
  FfnBwdGeom ge{};
  ge.XB = ffn_bwd_xblocks(a.M);
  ge.RC = ffn_bwd_chunks_for(a.M, FF);
  const int nrb16 = (a.M + 15) / 16;
  ge.per = (nrb16 + ge.RC - 1) / ge.RC;
  ge.NS = FF / kFbSlice;
  size_t lds = ffn_bwd_lds_bytes<T>(FF, a.g_bn != nullptr);
  const int role = cb.cj != nullptr ? ((cb.C + kCoeffThreads - 1) / kCoeffThreads) * cb.G : 0;
  if (role > 0 && sizeof(float) * (size_t)(kCoeffPass * cb.N) > lds) lds = sizeof(float) * (size_t)(kCoeffPass * cb.N);
  auto kern = ffn_bwd_kernel<T, FF>;
  static LdsSeen seen;
  allow_dynamic_lds(kern, lds, seen);
  // XCD-aware order: every 32-row block has its own X-role workgroup and a chunk is exactly two of them
  const int nblk = (a.M + kFbRows - 1) / kFbRows;
  ge.xcd = (ge.XB == nblk && ge.per == 4) ? 1 : 0;
  if (const char* e = getenv("FETA_FFN_BWD_XCD")) ge.xcd = ge.xcd && atoi(e) != 0;
  ge.wxcd = ge.xcd ? 0 : 1;
  if (const char* e = getenv("FETA_FFN_BWD_WXCD")) ge.wxcd = ge.wxcd && atoi(e) != 0;
  // chunk-wise X role (FETA_FFN_BWD_CXW=1; needs whole 32-row blocks per chunk and a whole number of walkers per chunk).
  // Measured at B = 16384 (round 3): the W-role order alone 684 us / 1.74 GB (launch order: 725 us / 3.12 GB; algorithmic
  // 0.94 GB), with the X role chunk-wise as well 829 us / 1.88 GB - walkers and slices do not stay in step, and the
  // strided X blocks had spread the tail better.  Off by default.
  ge.cxw = 0;
  if (const char* e = getenv("FETA_FFN_BWD_CXW")) {
    if (atoi(e) != 0 && !ge.xcd && ge.wxcd && ge.per % 2 == 0 && ge.XB >= ge.RC && ge.XB % ge.RC == 0) ge.cxw = ge.XB / ge.RC;
  }
  if (ge.cxw > 0) ge.wxcd = 0;
  const int grid = ge.xcd ? 8 * ((ge.RC + 7) / 8) * (2 + ge.NS)
                 : ge.cxw > 0 ? 8 * ((ge.RC + 7) / 8) * (ge.cxw + ge.NS)
                              : ge.XB + (ge.wxcd ? 8 * ((ge.RC + 7) / 8) * ge.NS : ge.RC * ge.NS);
  ge.main_grid = grid;
  ge.role = role;
  ge.role_lead = 8 * ((role + 7) / 8);
  if (const char* e = getenv("FETA_COEFF_ROLE_FIRST")) if (atoi(e) == 0) ge.role_lead = 0;
  hipLaunchKernelGGL(kern, dim3(grid + (ge.role_lead > 0 ? ge.role_lead : role)), dim3(kRowThreads), lds, stream, a, ge, cb);
  return check_launch("feta_ffn_bwd");
}

}  // namespace feta

using namespace feta;

extern "C" int feta_ffn_bwd_supported(int d_model, int ff) { return (d_model == kFbD && (ff == 64 || ff == 128)) ? 1 : 0; }

extern "C" int feta_ffn_bwd_blocks(int M) { return ffn_bwd_xblocks(M); }
extern "C" int feta_ffn_bwd_chunks(int M, int ff) { return (M < 1 || !feta_ffn_bwd_supported(kFbD, ff)) ? 0 : ffn_bwd_chunks_for(M, ff); }

extern "C" int feta_ffn_bwd(const feta_ffn_grad* d, feta_stream_t stream) { return feta_ffn_bwd_coeff(d, nullptr, stream); }

extern "C" int feta_ffn_bwd_coeff(const feta_ffn_grad* d, const feta_coeff_bwd_role* c, feta_stream_t stream) {
  FETA_REQUIRE(d != nullptr, "ffn_bwd: null descriptor");
  CoeffBwdRole cb{};
  if (c != nullptr) {
    FETA_REQUIRE(c->cj && c->n_real && c->s && c->gcn_bias && c->dpooled && c->partial, "ffn_bwd_coeff: null pointer");
    FETA_REQUIRE(c->B > 0 && c->H > 0 && c->C > 0 && c->N > 0, "ffn_bwd_coeff: empty shape");
    cb = CoeffBwdRole{c->cj, c->n_real, c->s, c->gcn_bias, c->dpooled, c->partial, c->B, c->N, c->H, c->C,
                      feta_coeff_bwd_groups(c->B, c->H)};
  }
  const FfnGradArgs& a = *d;
  FETA_REQUIRE(a.dy && a.h && a.w2 && a.w1 && a.x && a.dx && a.partial && a.M > 0, "ffn_bwd: null pointer / empty");
  FETA_REQUIRE(feta_ffn_bwd_supported(kFbD, a.FF), "ffn_bwd: dim_feedforward %d not in {64,128}", a.FF);
  FETA_REQUIRE(!a.g_y || a.g_ln_gamma || (a.g_bn && (a.g_sum || a.g_fin)), "ffn_bwd: g_y needs g_bn and g_sum | g_fin, or g_ln_gamma");
  FETA_REQUIRE(!a.g_ln_gamma || (a.g_y && !a.g_bn && !a.g_sum && !a.g_fin), "ffn_bwd: g_ln_gamma needs g_y and excludes g_bn / g_sum / g_fin");
  FETA_REQUIRE(!a.x_ln_gamma || (a.x_ln_beta && !a.x_bn), "ffn_bwd: x_ln_gamma needs x_ln_beta and excludes x_bn");
  FETA_REQUIRE(!a.g_bn || a.g_y, "ffn_bwd: g_bn needs g_y");
  FETA_REQUIRE(!a.g_sum || a.Gs > 0, "ffn_bwd: g_sum needs Gs");
  FETA_REQUIRE(!a.sum_out || a.x_bn, "ffn_bwd: sum_out needs x_bn (the BatchNorm that produced x)");
  FETA_REQUIRE(aligned16(a.dy_b) && aligned16(a.dy) && aligned16(a.h) && aligned16(a.x) && aligned16(a.dx) && aligned16(a.g_y) &&
               aligned16(a.x_bn) && aligned16(a.g_sum), "ffn_bwd: pointers must be 16-byte aligned");
  FETA_REQUIRE(a.dtype == FETA_F32 || a.dtype == FETA_BF16, "ffn_bwd: dtype %d", a.dtype);
  if (a.dtype == FETA_BF16) {
    if (a.FF == 64) return launch_ffn_bwd<bf16_t, 64>(a, cb, (hipStream_t)stream);
    return launch_ffn_bwd<bf16_t, 128>(a, cb, (hipStream_t)stream);
  }
  if (a.FF == 64) return launch_ffn_bwd<float, 64>(a, cb, (hipStream_t)stream);
  return launch_ffn_bwd<float, 128>(a, cb, (hipStream_t)stream);
}
