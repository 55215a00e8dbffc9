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
  int main_grid;   // workgroups of the two roles above; beyond: the coefficient generator's backward (feta_coeff.h)
};

inline int ffn_bwd_xblocks(int M) {
  int cap = 512;
  if (const char* e = getenv("FETA_FFN_MAX_GRID")) cap = atoi(e) > 0 ? atoi(e) : cap;
  const int nblk = (M + kFbRows - 1) / kFbRows;
  return nblk < cap ? nblk : cap;
}

// g2(row, o..o+3) from dy (and y2 when the BatchNorm backward is folded in); gv = [5][64]: scale, mean, rstd, m1, m2
__device__ __forceinline__ float4 g2_of(const float4& dv, const float4& yv, const float* gv, int o, bool gbn) {
  float v[4] = {dv.x, dv.y, dv.z, dv.w};
  if (gbn) {
    const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float xh = (yy[s] - gv[kFbD + o + s]) * gv[2 * kFbD + o + s];
      v[s] = gv[o + s] * (v[s] - gv[3 * kFbD + o + s] - xh * gv[4 * kFbD + o + s]);
    }
  }
  return make_float4(v[0], v[1], v[2], v[3]);
}

template <int FF>
__global__ __launch_bounds__(kRowThreads) void ffn_bwd_kernel(FfnGradArgs a, FfnBwdGeom ge, CoeffBwdRole cb) {
  constexpr int D = kFbD, NJ2 = FF / 16;
  if ((int)blockIdx.x >= ge.main_grid) {
    // trailing workgroups: the backward of the coefficient generator (feta_coeff.h) - it depends on the filter stage
    // only, and this is the first launch of the layer stack's backward
    const int r = (int)blockIdx.x - ge.main_grid, nbx = (cb.C + kCoeffThreads - 1) / kCoeffThreads;
    coeff_bwd_body(cb.cj, cb.n_real, cb.s, cb.gbias, cb.dpooled, cb.partial, cb.B, cb.N, cb.H, cb.C, cb.G, r % nbx, r / nbx);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lq = lane & 15, g = lane >> 4;
  // Which piece of work this workgroup is.  A 64-row chunk is touched by six workgroups (its two X-role blocks and the
  // four hidden-unit slices of the W role), and workgroups are dealt round-robin to the 8 XCDs, each with its own L2:
  // in launch order (X blocks, then W slices) the six sat on up to six XCDs and the chunk's rows of dy, y2, y1 left HBM
  // up to six times (2.3x the algorithmic bytes by the counters).  With ge.xcd the grid is read as
  // (chunk group, item, XCD): all six items of chunk 8 * group + XCD run on that XCD.
  int xblk, wbi;
  if (ge.xcd) {
    const int xcd = (int)blockIdx.x & 7, v = (int)blockIdx.x >> 3, items = 2 + ge.NS;
    const int item = v % items, rc = (v / items) * 8 + xcd;
    if (rc >= ge.RC) return;
    xblk = item < 2 ? 2 * rc + item : -1;
    wbi = item < 2 ? -1 : rc * ge.NS + (item - 2);
    if (item < 2 && xblk >= ge.XB) return;
  } else {
    xblk = (int)blockIdx.x < ge.XB ? (int)blockIdx.x : -1;
    wbi = (int)blockIdx.x - ge.XB;
  }
  float* gv = feta_lds;   // [5][64]
  float* after = gv;
  const bool gbn = a.g_y != nullptr;
  if (gbn) {
    float* scr = gv + 5 * D;
    after = scr;
    const int cpre = min(tid, D - 1);
    const float bn_scale = a.g_bn[cpre], bn_mean = a.g_bn[2 * D + cpre], bn_rstd = a.g_bn[3 * D + cpre];
    if (a.g_sum != nullptr) {
      reduce_partials(a.g_sum, a.Gs, D, scr + 2 * D, scr);
      for (int c = tid; c < D; c += kRowThreads) {
        gv[3 * D + c] = scr[c] / (float)a.M;
        gv[4 * D + c] = scr[D + c] / (float)a.M;
        if (blockIdx.x == 0) {
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

  if (xblk >= 0) {
    // ================================ X role: dh tile -> dx1 ================================================
    constexpr int GP = D + 4, DP = FF + 4, CT2 = FF / 64;   // CT2 column tiles of dh per wave
    float* gt = after;              // [32][GP]  g2
    float* dht = gt + kFbRows * GP; // [32][DP]  dh
    // weight slices of this wave, for the whole launch
    float wA2[CT2][4][4];           // W2[o = 16j+4g+s][c = 16 (w CT2 + t) + lq]
#pragma unroll
    for (int t = 0; t < CT2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) wA2[t][j][s] = a.w2[(int64_t)(16 * j + 4 * g + s) * FF + 16 * (w * CT2 + t) + lq];
    float wA1[NJ2][4];              // W1[c = 16j+4g+s][k = 16w + lq]
#pragma unroll
    for (int j = 0; j < NJ2; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) wA1[j][s] = a.w1[(int64_t)(16 * j + 4 * g + s) * D + 16 * w + lq];
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
    for (int blk = xblk; blk < nblk; blk += ge.XB) {
      const int r0 = blk * kFbRows;
      if (blk != xblk) __syncthreads();   // the tiles of the previous block have been consumed
      // requests of the block: gradient tile (2 items per thread), relu operands, residual rows of the sums
      float4 dv[2], yv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * kRowThreads, rr = idx >> 4, c4 = idx & 15;
        const int64_t off = (int64_t)min(r0 + rr, row_last) * D + 4 * c4;
        dv[u] = *reinterpret_cast<const float4*>(a.dy + off);
        if (a.dy_b != nullptr) {   // the gradient arrives in two parts (feta_attn_block_bwd, SPLIT form)
          const float4 d2 = *reinterpret_cast<const float4*>(a.dy_b + off);
          dv[u] = make_float4(dv[u].x + d2.x, dv[u].y + d2.y, dv[u].z + d2.z, dv[u].w + d2.w);
        }
        yv[u] = *reinterpret_cast<const float4*>((gbn ? a.g_y : a.dy) + off);
      }
      float4 hv[2][CT2], sy[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int64_t rowc = min(r0 + 16 * rt + lq, row_last);
#pragma unroll
        for (int t = 0; t < CT2; ++t)
          hv[rt][t] = *reinterpret_cast<const float4*>(a.h + rowc * FF + 16 * (w * CT2 + t) + 4 * g);
        sy[rt] = *reinterpret_cast<const float4*>(a.x + rowc * D + 16 * w + 4 * g);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * kRowThreads, rr = idx >> 4, c4 = idx & 15;
        const float4 v = g2_of(dv[u], yv[u], gv, 4 * c4, gbn);
        const bool ok = r0 + rr < a.M;
        *reinterpret_cast<float4*>(gt + rr * GP + 4 * c4) =
            make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
      }
      __syncthreads();
      // dh^T tiles (c = 16 ct + 4g + r, row = 16 rt + lq) = sum_o W2[o][c] g2[row][o], masked by h > 0
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        Feat<D> gf;
        load_row<D>(gf, gt + (16 * rt + lq) * GP, g);
#pragma unroll
        for (int t = 0; t < CT2; ++t) {
          f32x4 acc = zero4();
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma16(wA2[t][j][s], gf.f[j][s], acc);
          const float4 hh = hv[rt][t];
          *reinterpret_cast<float4*>(dht + (16 * rt + lq) * DP + 16 * (w * CT2 + t) + 4 * g) =
              make_float4(hh.x > 0.0f ? acc[0] : 0.0f, hh.y > 0.0f ? acc[1] : 0.0f, hh.z > 0.0f ? acc[2] : 0.0f,
                          hh.w > 0.0f ? acc[3] : 0.0f);
        }
      }
      __syncthreads();
      // dx1^T tiles (k = 16w + 4g + r, row) = sum_c W1[c][k] dh[row][c]  + g2[row][k]
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        Feat<FF> df;
        load_row<FF>(df, dht + (16 * rt + lq) * DP, g);
        f32x4 acc = zero4();
#pragma unroll
        for (int j = 0; j < NJ2; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma16(wA1[j][s], df.f[j][s], acc);
        const float4 res = *reinterpret_cast<const float4*>(gt + (16 * rt + lq) * GP + 16 * w + 4 * g);
        const float v[4] = {acc[0] + res.x, acc[1] + res.y, acc[2] + res.z, acc[3] + res.w};
        const int row = r0 + 16 * rt + lq;
        const bool rok = row < a.M;
        if (rok) *reinterpret_cast<float4*>(a.dx + (int64_t)row * D + 16 * w + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
        if (want_sums) {
          const float yy[4] = {sy[rt].x, sy[rt].y, sy[rt].z, sy[rt].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xh = (yy[r] - mean1[r]) * rstd1[r];
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
  constexpr int GPW = D + 16, HP = kFbSlice + 16;
  const int bi = wbi;
  const int si = bi % ge.NS, rc = bi / ge.NS;
  const int c0 = kFbSlice * si;
  float* gt = after;             // [64][GPW] g2
  float* xt = gt + 64 * GPW;     // [64][GPW] x1 (seen through its BatchNorm)
  float* hs = xt + 64 * GPW;     // [64][HP]  h slice
  float* dhs = hs + 64 * HP;     // [64][HP]  dh slice
  float wA2[2][4][4];            // W2[o = 16j+4g+s][c = c0 + 16t + lq]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) wA2[t][j][s] = a.w2[(int64_t)(16 * j + 4 * g + s) * FF + c0 + 16 * t + lq];
  const int row_lo = rc * ge.per * 16, row_hi = min((rc + 1) * ge.per * 16, a.M);
  const int row_last = max(row_hi - 1, 0);
  f32x4 aW2[2] = {zero4(), zero4()}, aW1[2] = {zero4(), zero4()};
  float db2 = 0.0f, db1[2] = {0.0f, 0.0f};
  const bool xbn = a.x_bn != nullptr;
  for (int r0 = row_lo; r0 < row_hi; r0 += 64) {
    if (r0 > row_lo) __syncthreads();
    // stage g2 [64 x 64], x1 [64 x 64] (4 items per thread each) and the h slice [64 x 32] (2 per thread): every
    // load of the pass is issued before the first LDS store
    float4 dv[4], yv[4], xv[4], hq[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + u * kRowThreads, rr = idx >> 4, c4 = idx & 15;
      const int64_t off = (int64_t)min(r0 + rr, row_last) * D + 4 * c4;
      dv[u] = *reinterpret_cast<const float4*>(a.dy + off);
      if (a.dy_b != nullptr) {
        const float4 d2 = *reinterpret_cast<const float4*>(a.dy_b + off);
        dv[u] = make_float4(dv[u].x + d2.x, dv[u].y + d2.y, dv[u].z + d2.z, dv[u].w + d2.w);
      }
      yv[u] = *reinterpret_cast<const float4*>((gbn ? a.g_y : a.dy) + off);
      xv[u] = *reinterpret_cast<const float4*>(a.x + off);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * kRowThreads, rr = idx >> 3, c4 = idx & 7;
      hq[u] = *reinterpret_cast<const float4*>(a.h + (int64_t)min(r0 + rr, row_last) * FF + c0 + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + u * kRowThreads, rr = idx >> 4, c4 = idx & 15;
      const bool ok = r0 + rr < row_hi;
      const float4 v = g2_of(dv[u], yv[u], gv, 4 * c4, gbn);
      *reinterpret_cast<float4*>(gt + rr * GPW + 4 * c4) =
          make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
      float4 x = xv[u];
      if (xbn) {
        const float4 sc = *reinterpret_cast<const float4*>(a.x_bn + 4 * c4);
        const float4 sh = *reinterpret_cast<const float4*>(a.x_bn + D + 4 * c4);
        x = make_float4(x.x * sc.x + sh.x, x.y * sc.y + sh.y, x.z * sc.z + sh.z, x.w * sc.w + sh.w);
      }
      *reinterpret_cast<float4*>(xt + rr * GPW + 4 * c4) =
          make_float4(ok ? x.x : 0.0f, ok ? x.y : 0.0f, ok ? x.z : 0.0f, ok ? x.w : 0.0f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * kRowThreads, rr = idx >> 3, c4 = idx & 7;
      const bool ok = r0 + rr < row_hi;
      *reinterpret_cast<float4*>(hs + rr * HP + 4 * c4) =
          make_float4(ok ? hq[u].x : 0.0f, ok ? hq[u].y : 0.0f, ok ? hq[u].z : 0.0f, ok ? hq[u].w : 0.0f);
    }
    __syncthreads();
    // this wave's row tile of the dh slice: (c = c0 + 16t + 4g + r, row = 16w + lq)
    {
      Feat<D> gf;
      load_row<D>(gf, gt + (16 * w + lq) * GPW, g);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = zero4();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma16(wA2[t][j][s], gf.f[j][s], acc);
        const float4 hh = *reinterpret_cast<const float4*>(hs + (16 * w + lq) * HP + 16 * t + 4 * g);
        *reinterpret_cast<float4*>(dhs + (16 * w + lq) * HP + 16 * t + 4 * g) =
            make_float4(hh.x > 0.0f ? acc[0] : 0.0f, hh.y > 0.0f ? acc[1] : 0.0f, hh.z > 0.0f ? acc[2] : 0.0f,
                        hh.w > 0.0f ? acc[3] : 0.0f);
      }
    }
    __syncthreads();
    // dW2[o = 16w + 4g' + r][c = c0 + 16ct + lq] += g2[row][o] h[row][c];  dW1[c][k = 16w + lq] += dh[row][c] x1[row][k]
#pragma unroll 4
    for (int st = 0; st < 16; ++st) {
      const int rr = 4 * st + g;
      const float ga = gt[rr * GPW + 16 * w + lq];
      const float xb = xt[rr * GPW + 16 * w + lq];
      db2 += ga;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const float hb = hs[rr * HP + 16 * ct + lq];
        const float da = dhs[rr * HP + 16 * ct + lq];
        db1[ct] += da;
        aW2[ct] = mfma16(ga, hb, aW2[ct]);
        aW1[ct] = mfma16(da, xb, aW1[ct]);
      }
    }
  }
  float* p = a.partial + (int64_t)rc * (a.partial_ld > 0 ? (int64_t)a.partial_ld : (int64_t)(2 * D * FF + D + FF));
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
  db2 += shfl_xor(db2, 16);
  db2 += shfl_xor(db2, 32);
  if (g == 0 && si == 0) pb2[16 * w + lq] = db2;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    float s = db1[ct];
    s += shfl_xor(s, 16);
    s += shfl_xor(s, 32);
    if (g == 0 && w == 0) pb1[c0 + 16 * ct + lq] = s;
  }
}

extern int row_chunks(int M);   // rowwise.hip

template <int FF>
int launch_ffn_bwd(const FfnGradArgs& a, const CoeffBwdRole& cb, hipStream_t stream) {
  FfnBwdGeom ge{};
  ge.XB = ffn_bwd_xblocks(a.M);
  ge.RC = row_chunks(a.M);
  const int nrb16 = (a.M + 15) / 16;
  ge.per = (nrb16 + ge.RC - 1) / ge.RC;
  ge.NS = FF / kFbSlice;
  const size_t x_lds = kFbRows * (kFbD + 4) + kFbRows * (FF + 4);
  const size_t w_lds = 2 * 64 * (kFbD + 16) + 2 * 64 * (kFbSlice + 16);
  size_t floats = (x_lds > w_lds ? x_lds : w_lds) + (a.g_y ? 5 * kFbD + reduce_scratch_floats(kFbD) : 0);
  const int role = cb.cj != nullptr ? ((cb.C + kCoeffThreads - 1) / kCoeffThreads) * cb.G : 0;
  if (role > 0 && (size_t)(kCoeffPass * cb.N) > floats) floats = kCoeffPass * cb.N;
  const size_t lds = sizeof(float) * floats;
  auto kern = ffn_bwd_kernel<FF>;
  static LdsSeen seen;
  allow_dynamic_lds(kern, lds, seen);
  // XCD-aware order: every 32-row block has its own X-role workgroup and a chunk is exactly two of them
  const int nblk = (a.M + kFbRows - 1) / kFbRows;
  ge.xcd = (ge.XB == nblk && ge.per == 4) ? 1 : 0;
  if (const char* e = getenv("FETA_FFN_BWD_XCD")) ge.xcd = ge.xcd && atoi(e) != 0;
  const int grid = ge.xcd ? 8 * ((ge.RC + 7) / 8) * (2 + ge.NS) : ge.XB + ge.RC * ge.NS;
  ge.main_grid = grid;
  hipLaunchKernelGGL(kern, dim3(grid + role), dim3(kRowThreads), lds, stream, a, ge, cb);
  return check_launch("feta_ffn_bwd");
}

}  // namespace feta

using namespace feta;

extern "C" int feta_ffn_bwd_supported(int d_model, int ff) { return (d_model == kFbD && (ff == 64 || ff == 128)) ? 1 : 0; }

extern "C" int feta_ffn_bwd_blocks(int M) { return ffn_bwd_xblocks(M); }

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
  FETA_REQUIRE(!a.g_y || (a.g_bn && (a.g_sum || a.g_fin)), "ffn_bwd: g_y needs g_bn and g_sum | g_fin");
  FETA_REQUIRE(!a.g_sum || a.Gs > 0, "ffn_bwd: g_sum needs Gs");
  FETA_REQUIRE(!a.sum_out || a.x_bn, "ffn_bwd: sum_out needs x_bn (the BatchNorm that produced x)");
  FETA_REQUIRE(aligned16(a.dy_b) && aligned16(a.dy) && aligned16(a.h) && aligned16(a.x) && aligned16(a.dx) && aligned16(a.g_y) &&
               aligned16(a.x_bn) && aligned16(a.g_sum), "ffn_bwd: pointers must be 16-byte aligned");
  if (a.FF == 64) return launch_ffn_bwd<64>(a, cb, (hipStream_t)stream);
  return launch_ffn_bwd<128>(a, cb, (hipStream_t)stream);
}
