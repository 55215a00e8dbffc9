// Feed-forward half of DiffTransformerEncoderLayer (BatchNorm variant) as ONE launch:
//   x = BN1(y1);  h = relu(x W1^T + b1);  y2 = x + h W2^T + b2;  statistics of y2 for BN2
// (contract transformer/models.py:166-167; body per upstream GraphiT, README.md:129: linear2(relu(
// linear1(.))) with dim_feedforward = 2 d, residual, norm2).  Replaces two feta_rowlin_fwd_ex
// launches: the hidden activations never leave the registers between the two products - the
// accumulator tile of the first product (rows = hidden units) is exactly the B operand of the
// second one, which contracts over hidden units (feta_tiles.h: contraction over accumulator rows).
//
// Workgroup = 32 rows x 4 waves; wave (rt, hh) owns row tile rt and HALF of the hidden units, so a
// wave's dependent MFMA chain is 128 instead of 256 instructions (the problem is latency-bound at
// BASELINE batch sizes: 4736 rows).  The two halves exchange their partial y2 tiles through LDS and
// each finishes two of the four output tiles; the residual BN1(y1) is the wave's own x operand.
// h is still written to HBM: backward needs it (relu mask, dW2).
#include <cstdlib>

#include "feta_abi_common.h"
#include "feta_coeff.h"
#include "feta_ln.h"
#include "feta_lp.h"
#include "feta_rowops.h"

namespace feta {

typedef feta_ffn FfnArgs;  // include/feta_hip.h

constexpr int kFfnD = 64;
constexpr int kFfnRows = 32;  // rows per workgroup

// LDS bytes of a workgroup: W1 / W2 tiles of T, fp32 for everything else
template <class T>
__host__ __device__ inline int ffn_lds_bytes(int ff) {
  const int pad = Lp<T>::PAD;
  return (int)sizeof(T) * (ff * (kFfnD + pad) + kFfnD * (ff + pad))  // W1 [FF][64+pad], W2 [64][FF+pad]
         + 4 * (2 * kFfnD                                              // scale / shift of BN1
                + reduce_scratch_floats(kFfnD)                        // finalize scratch, later: partial-tile exchange + stats
                + 4 * 2 * 4 * 64);                                    // exchange [wave][2 tiles][4 regs][64 lanes]
}

// workgroups of a launch = partial rows of y_stats: every 32-row block up to 512 of them (two resident
// per CU), beyond that the workgroups loop.  FETA_FFN_MAX_GRID: tests force the loop.
inline int ffn_grid(int M) {
  int cap = 512;
  if (const char* e = getenv("FETA_FFN_MAX_GRID")) cap = atoi(e) > 0 ? atoi(e) : cap;
  const int nblk = (M + kFfnRows - 1) / kFfnRows;
  return nblk < cap ? nblk : cap;
}

#ifdef FETA_TIMING
__device__ unsigned long long feta_ffn_stamps[16];
#endif
#define FFN_STAMP(i) FETA_STAMP_TO(feta_ffn_stamps, i, blockIdx.x == 0 && threadIdx.x == 0)

// Workgroups beyond main_grid run the forward of the coefficient generator (feta_coeff.h), one (head, graph) block each:
// it depends on the attention matrix of the last layer only, so it shares the launch of that layer's feed-forward half.
// T: storage type of x, h, y and of the weight tiles in LDS (feta_lp.h); weights in HBM, biases, statistics: fp32.
template <class T, int FF>
__global__ __launch_bounds__(kRowThreads) void ffn_fwd_kernel(FfnArgs a, CoeffFwdRole cf, int main_grid) {
  typedef Lp<T> L;
  typedef typename L::Op Op;
  constexpr int D = kFfnD, P1 = D + L::PAD, P2 = FF + L::PAD, HT = FF / 32;  // HT hidden tiles per half
  if ((int)blockIdx.x >= main_grid) {
    coeff_fwd_body(cf.attn, cf.n_real, cf.s, cf.gbias, cf.cj, cf.pooled, cf.B, cf.N, cf.H, cf.C, coeff_fwd_stage(cf.N),
                   (int)blockIdx.x - main_grid);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lq = lane & 15, g = lane >> 4;
  const int rt = wv >> 1, hh = wv & 1;
  T* W1 = reinterpret_cast<T*>(lds_bytes());   // [FF][P1]
  T* W2 = W1 + FF * P1;                         // [64][P2]
  float* xss = reinterpret_cast<float*>(W2 + D * P2);   // [2][64]
  float* scr = xss + 2 * D;        // finalize scratch
  float* xch = scr + reduce_scratch_floats(D);  // [4 waves][2][4][64]
  const T* gx = reinterpret_cast<const T*>(a.x);
  T* gh = reinterpret_cast<T*>(a.h);
  T* gy = reinterpret_cast<T*>(a.y);
  FFN_STAMP(0);
  const int nblk = (a.M + kFfnRows - 1) / kFfnRows;
  // a workgroup stages the weights ONCE and walks its row blocks (blockIdx.x, + gridDim.x, ...): at large
  // batches the 64 KB of weights per 32 rows would otherwise be the largest stream of the kernel
  int row = blockIdx.x * kFfnRows + 16 * rt + lq;
  bool rok = row < a.M;
  int rowc = min(row, a.M - 1);

  // ---- requests: the wave's first x rows, biases, then the weights ----------------------------------------
  float xr[4][4];   // the wave's x row chunks (features 16j + 4g ..), fp32: operand of the first product AND the residual
  // ... as they arrive from memory (storage type): the rows of the NEXT row block are requested while the block at hand is
  // computed (a workgroup walks 4 blocks at config 5's batch with one or two waves per SIMD: nobody else hides the latency)
  Op xq[4];
  auto request_x = [&](int rowc_) {
#pragma unroll
    for (int j = 0; j < 4; ++j) xq[j] = L::ld(gx + (int64_t)rowc_ * D + 16 * j + 4 * g);
  };
  request_x(rowc);
  float4 b1v[HT], b2v[2], ksv[2];   // ksv: shift of the y statistics (feta_rowops.h) for this wave's two output tiles
#pragma unroll
  for (int t = 0; t < HT; ++t)
    b1v[t] = a.b1 != nullptr ? *reinterpret_cast<const float4*>(a.b1 + hh * (FF / 2) + 16 * t + 4 * g)
                             : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    b2v[t] = a.b2 != nullptr ? *reinterpret_cast<const float4*>(a.b2 + 16 * (2 * hh + t) + 4 * g)
                             : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    ksv[t] = (a.y_shift != nullptr && a.y_stats != nullptr)
                 ? *reinterpret_cast<const float4*>(a.y_shift + 16 * (2 * hh + t) + 4 * g)
                 : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  // LayerNorm of the OUTPUT rows in the epilogue (y_ln_out: norm2 of a LayerNorm stack whose consumer is not one of the
  // on-load kernels - graphs beyond 64 nodes, or the end of the stack): gamma / beta of this wave's two output tiles
  const bool y_ln = a.y_ln_out != nullptr;
  float4 lgv[2], lbv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    lgv[t] = y_ln ? *reinterpret_cast<const float4*>(a.y_ln_gamma + 16 * (2 * hh + t) + 4 * g) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    lbv[t] = y_ln ? *reinterpret_cast<const float4*>(a.y_ln_beta + 16 * (2 * hh + t) + 4 * g) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  // gamma, beta and the shift of the input BatchNorm travel with the first requests (three floats for each of the first D
  // threads): read after the reduction they were one more dependent round trip in the middle of the prologue
  float xg = 1.0f, xb = 0.0f, xk = 0.0f;
  if (a.x_stats != nullptr && tid < D) {
    xg = a.x_gamma[tid];
    xb = a.x_beta[tid];
    xk = partials_shift(a.x_stats, a.Gx, D, tid);
  }
  {
    constexpr int NV = 2 * FF * D / 4 / kRowThreads;  // float4 per thread: W1 then W2 (fp32 masters)
    float4 wv4[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + kRowThreads * i;
      wv4[i] = idx < FF * D / 4 ? reinterpret_cast<const float4*>(a.w1)[idx]
                                : reinterpret_cast<const float4*>(a.w2)[idx - FF * D / 4];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + kRowThreads * i;
      if (idx < FF * D / 4) {
        L::st4(W1 + (idx / (D / 4)) * P1 + 4 * (idx % (D / 4)), wv4[i].x, wv4[i].y, wv4[i].z, wv4[i].w);
      } else {
        const int j = idx - FF * D / 4;
        L::st4(W2 + (j / (FF / 4)) * P2 + 4 * (j % (FF / 4)), wv4[i].x, wv4[i].y, wv4[i].z, wv4[i].w);
      }
    }
  }
  FFN_STAMP(1);
  if (a.x_stats != nullptr) {
    // (requesting the partial sums together with the weights was measured SLOWER, 18.0 k vs 15.0 k cycles:
    // every workgroup asks for the same rows at the same moment, and the in-order return of loads parks the
    // weights behind that hot spot)
    reduce_partials_t<kRowThreads, 32>(a.x_stats, a.Gx, D, scr + 2 * D, scr);   // (256 rows in ONE batch: feta_rowops.h)
    for (int c = tid; c < D; c += kRowThreads) {
      float mean, var;
      bn_moments_k(xk, D, a.M, scr, c, mean, var);   // (c == tid: the loop runs once for the first D threads)
      const float rstd = rsqrtf(var + a.eps);
      const float scale = xg * rstd;
      const float shift = xb - mean * scale;
      xss[c] = scale;
      xss[D + c] = shift;
      if (blockIdx.x == 0) {
        a.x_bn_out[c] = scale;
        a.x_bn_out[D + c] = shift;
        a.x_bn_out[2 * D + c] = mean;
        a.x_bn_out[3 * D + c] = rstd;
        if (a.x_rmean != nullptr) {
          const float unbiased = a.M > 1 ? var * (float)a.M / (float)(a.M - 1) : var;
          a.x_rmean[c] = (1.0f - a.momentum) * a.x_rmean[c] + a.momentum * mean;
          a.x_rvar[c] = (1.0f - a.momentum) * a.x_rvar[c] + a.momentum * unbiased;
        }
        if (c == 0 && a.x_nbt != nullptr) *a.x_nbt += 1;
      }
    }
  } else if (a.x_ln_gamma != nullptr) {
    // LayerNorm on load (feta_ln.h): xss = gamma | beta, mean / rstd per row where the rows are loaded
    for (int c = tid; c < 2 * D; c += kRowThreads) xss[c] = c < D ? a.x_ln_gamma[c] : a.x_ln_beta[c - D];
  } else {
    for (int c = tid; c < 2 * D; c += kRowThreads) xss[c] = a.x_bn != nullptr ? a.x_bn[c] : (c < D ? 1.0f : 0.0f);
  }
  const bool x_ln = a.x_ln_gamma != nullptr;
  const float ln_eps = a.eps;
  __syncthreads();
  FFN_STAMP(2);
  const bool want_stats = a.y_stats != nullptr;
  float tot1[1] = {0.0f};   // thread tid < 128: one entry of the workgroup's [2][64] (sum, sum of squares)
  const int lane0 = lane;
  for (int blk = blockIdx.x; blk < nblk; blk += main_grid) {
  // (what a lane derives from its id is invariant in this loop and would be hoisted and held across the body; the id is
  // laundered once per row block - csrc/block_bwd.hip)
  int lane_l = lane0;
  FETA_OPAQUE_LANE(lane_l);
  const int lane = lane_l, tid = (wv << 6) | lane, lq = lane & 15, g = lane >> 4;
  if (blk != (int)blockIdx.x) {
    lds_barrier();   // exchange / reduction scratch of the previous row block consumed (LDS data only: loads stay in flight)
    row = blk * kFfnRows + 16 * rt + lq;
    rok = row < a.M;
    rowc = min(row, a.M - 1);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) xr[j][e] = L::get(xq[j], e);
  if (blk + main_grid < nblk)   // (wave-uniform)
    request_x(min((blk + main_grid) * kFfnRows + 16 * rt + lq, a.M - 1));
  RowOp<T, D> xf;
  if (x_ln) {
    // the row lq of this wave's tile is spread over the four lanes lq + 16 g (16 features each): mean and biased variance
    // in two passes over the registers, as F.layer_norm; the scale / shift below then are gamma / beta
    float s0 = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s0 += (xr[j][0] + xr[j][1]) + (xr[j][2] + xr[j][3]);
    s0 += shfl_xor(s0, 16);
    s0 += shfl_xor(s0, 32);
    const float mean = s0 * (1.0f / D);
    float q0 = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xr[j][e] -= mean;
        q0 += xr[j][e] * xr[j][e];
      }
    q0 += shfl_xor(q0, 16);
    q0 += shfl_xor(q0, 32);
    const float rstd = 1.0f / sqrtf(q0 * (1.0f / D) + ln_eps);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) xr[j][e] *= rstd;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 sc = *reinterpret_cast<const float4*>(xss + 16 * j + 4 * g);
    const float4 sh = *reinterpret_cast<const float4*>(xss + D + 16 * j + 4 * g);
    xr[j][0] = xr[j][0] * sc.x + sh.x;
    xr[j][1] = xr[j][1] * sc.y + sh.y;
    xr[j][2] = xr[j][2] * sc.z + sh.z;
    xr[j][3] = xr[j][3] * sc.w + sh.w;
    xf.o[j] = L::mk(xr[j][0], xr[j][1], xr[j][2], xr[j][3]);
  }

  // ---- h (this half of the hidden units) = relu(x W1^T + b1): (hidden 4g+r, row lq) ------------------
  Op ho[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const int o = hh * (FF / 2) + 16 * t;
    RowOp<T, D> wf;
    load_row_op<T, D>(wf, W1 + (o + lq) * P1, g);
    f32x4 v = dot_row_ops<T, D>(wf, xf, zero4());
    v[0] = fmaxf(v[0] + b1v[t].x, 0.0f);
    v[1] = fmaxf(v[1] + b1v[t].y, 0.0f);
    v[2] = fmaxf(v[2] + b1v[t].z, 0.0f);
    v[3] = fmaxf(v[3] + b1v[t].w, 0.0f);
    ho[t] = L::mk(v);   // the accumulator tile of the first product IS the B operand of the second
    if (rok) L::st4(gh + (int64_t)row * FF + o + 4 * g, v[0], v[1], v[2], v[3]);
  }
  FFN_STAMP(3);
  // ---- partial y2^T tiles over this half: (output 4g+r, row lq) --------------------------------------
  f32x4 yp[4];
#pragma unroll
  for (int t2 = 0; t2 < 4; ++t2) {
    f32x4 acc = zero4();
#pragma unroll
    for (int t = 0; t < HT; ++t)
      acc = L::mma(L::ld(W2 + (16 * t2 + lq) * P2 + hh * (FF / 2) + 16 * t + 4 * g), ho[t], acc);
    yp[t2] = acc;
  }
  FFN_STAMP(4);
  // ---- exchange: this wave finishes output tiles 2hh, 2hh+1; the partner gets the other two -----------
  {
    float* mine = xch + wv * (2 * 4 * 64);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)   // (selects, not a runtime index: that would put yp in private memory)
        mine[(t * 4 + r) * 64 + lane] = hh ? yp[t][r] : yp[2 + t][r];
  }
  lds_barrier();   // (the partial tiles live in LDS; the next block's rows keep travelling)
  const float* theirs = xch + (wv ^ 1) * (2 * 4 * 64);
  float* red = scr;  // [2 row tiles][2][64] column sums, reduced below
  float vv[2][4];    // this wave's 32 columns of row lq (kept for the LayerNorm epilogue)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int t2 = 2 * hh + t, o2 = 16 * t2 + 4 * g;
    const float bb[4] = {b2v[t].x, b2v[t].y, b2v[t].z, b2v[t].w};
    float* v = vv[t];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float own = hh ? yp[2 + t][r] : yp[t][r];
      // (the empty asm makes both candidates plain register values: a select between two LOADS of xr is
      // folded into one load at a selected address, which puts xr in private memory)
      float x_hi = xr[2 + t][r], x_lo = xr[t][r];
      asm volatile("" : "+v"(x_hi), "+v"(x_lo));
      const float res = hh ? x_hi : x_lo;
      v[r] = own + theirs[(t * 4 + r) * 64 + lane] + bb[r] + res;
    }
    if (rok) {
      if (a.y_f32) *reinterpret_cast<float4*>(a.y + (int64_t)row * D + o2) = make_float4(v[0], v[1], v[2], v[3]);
      else L::st4(gy + (int64_t)row * D + o2, v[0], v[1], v[2], v[3]);
    }
    if (want_stats) {
      const float kk[4] = {ksv[t].x, ksv[t].y, ksv[t].z, ksv[t].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x1 = rok ? v[r] - kk[r] : 0.0f;
        const float s1 = row16_sum(x1), s2 = row16_sum(x1 * x1);
        if (lq == 0) {
          red[(rt * 2 + 0) * D + o2 + r] = s1;
          red[(rt * 2 + 1) * D + o2 + r] = s2;
        }
      }
    }
  }
  if (want_stats) {
    lds_barrier();
    if (tid < 2 * D) tot1[0] += red[tid] + red[2 * D + tid];
  }
  if (y_ln) {
    // out = LayerNorm(y row) * gamma + beta: the row lq of tile rt is spread over the four lane groups of this wave (32
    // columns) and of its partner (the other 32) - two passes (mean, centred squares) as F.layer_norm, each one shuffle
    // pair and one hand-over through LDS (scr: no column statistics in a LayerNorm stack)
    float* lnx = scr;   // [2 passes][4 waves][16 rows]
    float s1 = ((vv[0][0] + vv[0][1]) + (vv[0][2] + vv[0][3])) + ((vv[1][0] + vv[1][1]) + (vv[1][2] + vv[1][3]));
    s1 += shfl_xor(s1, 16);
    s1 += shfl_xor(s1, 32);
    if (g == 0) lnx[wv * 16 + lq] = s1;
    lds_barrier();
    const float mean = (s1 + lnx[(wv ^ 1) * 16 + lq]) * (1.0f / (float)D);
    float s2 = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        vv[t][r] -= mean;
        s2 += vv[t][r] * vv[t][r];
      }
    s2 += shfl_xor(s2, 16);
    s2 += shfl_xor(s2, 32);
    if (g == 0) lnx[64 + wv * 16 + lq] = s2;
    lds_barrier();
    const float rstd = rsqrtf((s2 + lnx[64 + (wv ^ 1) * 16 + lq]) * (1.0f / (float)D) + a.y_ln_eps);
    if (rok) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int o2 = 16 * (2 * hh + t) + 4 * g;
        const float o0 = vv[t][0] * rstd * lgv[t].x + lbv[t].x, o1 = vv[t][1] * rstd * lgv[t].y + lbv[t].y;
        const float o2v = vv[t][2] * rstd * lgv[t].z + lbv[t].z, o3 = vv[t][3] * rstd * lgv[t].w + lbv[t].w;
        if (a.y_ln_f32) *reinterpret_cast<float4*>(a.y_ln_out + (int64_t)row * D + o2) = make_float4(o0, o1, o2v, o3);
        else L::st4(reinterpret_cast<T*>(a.y_ln_out) + (int64_t)row * D + o2, o0, o1, o2v, o3);
      }
    }
  }
  FFN_STAMP(5);
  }  // row blocks of this workgroup
  if (want_stats && tid < 2 * D) a.y_stats[(int64_t)blockIdx.x * 2 * D + tid] = tot1[0];
  if (want_stats && blockIdx.x == 0 && tid < D)   // row main_grid: the shift these sums are relative to
    a.y_stats[(int64_t)main_grid * 2 * D + tid] = a.y_shift != nullptr ? a.y_shift[tid] : 0.0f;
}

template <class T, int FF>
int launch_ffn_fwd(const FfnArgs& a, const CoeffFwdRole& cf, hipStream_t stream) {
  size_t lds = ffn_lds_bytes<T>(FF);
  const int role = cf.attn != nullptr ? cf.B * cf.H : 0;
  if (role > 0 && sizeof(float) * (size_t)coeff_fwd_lds_floats(cf.N) > lds) lds = sizeof(float) * coeff_fwd_lds_floats(cf.N);
  auto kern = ffn_fwd_kernel<T, FF>;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  const int grid = ffn_grid(a.M);
  hipLaunchKernelGGL(kern, dim3(grid + role), dim3(kRowThreads), lds, stream, a, cf, grid);
  return check_launch("feta_ffn_fwd");
}

template <class T>
int dispatch_ffn_fwd(const FfnArgs& a, const CoeffFwdRole& cf, hipStream_t stream) {
  switch (a.FF) {
    case 64: return launch_ffn_fwd<T, 64>(a, cf, stream);
    case 128: return launch_ffn_fwd<T, 128>(a, cf, stream);
    default: return launch_ffn_fwd<T, 256>(a, cf, stream);
  }
}

}  // namespace feta

using namespace feta;

#ifdef FETA_TIMING
extern "C" int feta_debug_ffn_stamps(unsigned long long* out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(feta_ffn_stamps), sizeof(unsigned long long) * 16);
}
#endif

extern "C" int feta_ffn_supported(int d_model, int ff) {
  return (d_model == kFfnD && (ff == 64 || ff == 128 || ff == 256)) ? 1 : 0;
}

extern "C" int feta_ffn_blocks(int M) { return ffn_grid(M); }

extern "C" int feta_ffn_fwd(const feta_ffn* d, feta_stream_t stream) { return feta_ffn_fwd_coeff(d, nullptr, stream); }

extern "C" int feta_ffn_fwd_coeff(const feta_ffn* d, const feta_coeff_fwd_role* c, feta_stream_t stream) {
  FETA_REQUIRE(d != nullptr, "ffn_fwd: null descriptor");
  CoeffFwdRole cf{};
  if (c != nullptr) {
    FETA_REQUIRE(c->attn && c->n_real && c->s && c->gcn_bias && c->cj && c->pooled, "ffn_fwd_coeff: null pointer");
    FETA_REQUIRE(c->B > 0 && c->H > 0 && c->C > 0 && c->N > 0 && c->N <= kCoeffThreads,
                 "ffn_fwd_coeff: need 0 < N <= %d (got %d)", kCoeffThreads, c->N);
    cf = CoeffFwdRole{c->attn, c->n_real, c->s, c->gcn_bias, c->cj, c->pooled, c->B, c->N, c->H, c->C};
  }
  const FfnArgs& a = *d;
  FETA_REQUIRE(a.x && a.w1 && a.w2 && a.h && a.y && a.M > 0, "ffn_fwd: null pointer / empty");
  FETA_REQUIRE(feta_ffn_supported(kFfnD, a.FF), "ffn_fwd: dim_feedforward %d not in {64,128,256}", a.FF);
  FETA_REQUIRE(a.x_stats == nullptr || (a.x_gamma && a.x_beta && a.x_bn_out && a.Gx > 0),
               "ffn_fwd: x_stats needs x_gamma, x_beta, x_bn_out, Gx");
  FETA_REQUIRE(a.x_ln_gamma == nullptr || (a.x_ln_beta != nullptr && a.x_stats == nullptr && a.x_bn == nullptr),
               "ffn_fwd: x_ln_gamma needs x_ln_beta and excludes x_bn / x_stats");
  FETA_REQUIRE(aligned16(a.x) && aligned16(a.w1) && aligned16(a.w2) && aligned16(a.h) && aligned16(a.y) &&
                   aligned16(a.b1) && aligned16(a.b2) && aligned16(a.x_stats) && aligned16(a.y_stats),
               "ffn_fwd: tensors must be 16-byte aligned");
  FETA_REQUIRE(a.y_ln_out == nullptr || (a.y_ln_gamma && a.y_ln_beta && a.y_stats == nullptr && aligned16(a.y_ln_out)),
               "ffn_fwd: y_ln_out needs y_ln_gamma, y_ln_beta and excludes y_stats");
  FETA_REQUIRE(a.dtype == FETA_F32 || a.dtype == FETA_BF16, "ffn_fwd: dtype %d", a.dtype);
  if (a.dtype == FETA_BF16) return dispatch_ffn_fwd<bf16_t>(a, cf, (hipStream_t)stream);
  return dispatch_ffn_fwd<float>(a, cf, (hipStream_t)stream);
}
