// Backward of the attention sub-block of one encoder layer as ONE launch (d = 64 = 4 heads x 16, N <= 64):
//   g1  = BatchNorm-1 backward of the incoming gradient (or the gradient itself: LayerNorm stack)   [N, 64]
//   dconcat = (degree * g1) W_out (+ the filter branch's gradient into out_each_head)               [N, 64]
//   dq | dk | dv = attention backward (arithmetic of attn_bwd_graph_kernel, attn.hip)                [N, 192]
//   dx  = dqkv W_in + g1   (+ the partial sums the previous layer's BatchNorm-2 backward needs)      [N, 64]
//   dW_out = (degree * g1)^T out, db_out;  dW_in = dqkv^T x0, db_in      (one partial row per workgroup)
// It replaces feta_rowlin_bwd_ex (out_proj) -> feta_attn_bwd -> feta_rowlin_bwd_ex (in_proj) of
// DiffTransformerEncoderLayer's backward (contract transformer/models.py:166-167,179,244; body per upstream
// GraphiT, README.md:129): three launches of ~4.5 us floor each, two of which re-stage what the third produced.
// dconcat and dqkv never reach HBM.
//
// One workgroup per graph, 8 waves = (head, role) exactly as attn_bwd_graph_kernel; the row-wise products use
// the same decomposition: a wave owns the 16 output columns of "its" head index and every other row tile
// (role = parity), with its weight column slice in registers (feta_rowlin_bwd's dX role).  Every graph writes its
// own partial row of the weight gradients (accumulators that outlive a graph cost ~50 registers through the
// attention phase: the kernel spilled), so the launch takes batches of up to kBbMaxGrid graphs; larger batches keep
// the three-launch form, which is the better shape there anyway (feta_rowlin_bwd's row chunks fill the chip).
//
// SPLIT form (dx_b given): TWO workgroups per graph, one per pair of heads - at the BASELINE batch one workgroup per
// graph leaves half of the 256 CUs idle and the kernel is MFMA-bound per CU.  Heads are independent up to dx, which
// contracts over all of dqkv: each workgroup writes the part of dx its heads contribute (dx: pair 0, with the residual
// g1; dx_b: pair 1) and the consumer adds the two on load (feta_ffn_bwd's dy_b); the partial sums of the previous
// BatchNorm's backward are linear in dx, so they are emitted per workgroup as well; dW_out columns / dW_in rows of the
// two pairs are disjoint parts of the graph's partial row.
#include <cstdlib>

#include "feta_abi_common.h"
#include "feta_colsum.h"
#include "feta_ln.h"
#include "feta_lp.h"
#include "feta_rowops.h"

namespace feta {

typedef feta_attn_block_grad BwdArgs;  // include/feta_hip.h

constexpr int kBbD = 64, kBbH = 4, kBbDH = 16;
constexpr int kBbP = kBbD + 4;   // pitch of a staged 64-float row
constexpr int kBbThreads = 512;
constexpr int kBbMaxGrid = 256;

// LDS bytes of a workgroup: seven [NR][64 + pad] tiles of T, fp32 for everything else
template <class T>
__host__ __device__ inline int block_bwd_lds_bytes(int nt, bool gbn, bool gln = false) {
  const int nr = 16 * nt, P = kBbD + Lp<T>::PAD;
  return (int)sizeof(T) * 7 * nr * P    // q, k, v (later dq, dk, dv), dconcat, out, g1, x0
         + 4 * (nr * (nr + 1)           // pe
                + kBbH * nr * 2 + nr    // softmax statistics, row scale
                + 8 * kBbD + 8 * 2 * 16 // fp32 column sums of the bias gradients: db_out per wave, dq | dk | dv per wave
                + (gbn ? 5 * kBbD + reduce_scratch_floats(kBbD, 512) : 0)
                + (gln ? 8 * 2 * kBbD : 0));   // LayerNorm stack: column sums of (dy xhat1, dy) per wave
}

#ifdef FETA_TIMING
__device__ unsigned long long feta_bbwd_stamps[4 * 8 * 8];   // FETA_RT_STAMP (feta_rowops.h), tools/block_timing.py
__device__ unsigned int feta_bbwd_launch;
#endif
#define BB_STAMP(i) FETA_RT_STAMP(feta_bbwd_stamps, feta_bbwd_launch, i)

// One element of a workgroup's partial row: the first graph it walks stores, every later one adds.  The add is a
// no-return float atomic - not for atomicity (only this thread ever touches the element, in program order) but because
// it has no result: a load-add-store would give the scheduler 40 independent loads per lane to hoist above the
// attention phase (the kernel spilled 0.5 - 1 KB per lane that way).
__device__ __forceinline__ void acc_to(float* p, float v, bool first) {
  if (first) *p = v;
  else atomicAdd(p, v);
}

// T: storage type of dy, y1, qkv, out, dout2, pe, x0, dx, dx_b and of the LDS tiles (feta_lp.h); weights, BatchNorm
// parameter blocks and partial sums, softmax statistics, row scale and the weight-gradient partial rows: fp32.
// LOOP: more graphs than workgroups - a separate instantiation, because the graph loop costs registers (the single-graph
// forms are spill-free; the compiler treats everything invariant in the loop as hoistable).
template <class T, int NT, bool SPLIT, bool LOOP>
__global__ __launch_bounds__(kBbThreads) void attn_block_bwd_kernel(BwdArgs a, ColsumPlan sums, int main_grid) {
  // Workgroups beyond main_grid reduce column sums (feta_colsum.h): the LAST launch of a stack's backward runs one workgroup
  // per graph - half the chip at the BASELINE batch - while every split-K partial of the layers behind it, and of this
  // layer's feed-forward half, is already complete; reduced here, the stack's final reduction launch is left with this
  // launch's own columns (40 MB -> 8.5 MB at the BASELINE batch)
  if ((int)blockIdx.x >= main_grid) {
    colsum_role<kBbThreads>(sums, (int)blockIdx.x - main_grid);
    return;
  }
  typedef Lp<T> L;
  typedef typename L::Op Op;
  typedef typename L::Vec Vec;
  constexpr int D = kBbD, DH = kBbDH, H = kBbH, P = kBbD + L::PAD, NR = 16 * NT, PEP = NR + 1;
  constexpr int VEC = L::VEC, RV = D / VEC;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lq = lane & 15, g = lane >> 4;
  // pair of heads of this workgroup: workgroups b and b + B share a graph - and, workgroups being dealt round-robin to
  // the 8 XCDs, an L2 when B is a multiple of 8 (adjacent workgroups never do: both would fetch the graph's tiles from
  // HBM, 1.5x the algorithmic bytes by the counters)
  const int hp = SPLIT ? ((int)blockIdx.x >= a.B ? 1 : 0) : 0;
  // attention part: wave = (head, role) - and, SPLIT, the parity of the tiles it walks
  const int h = SPLIT ? 2 * hp + (wv & 1) : (wv & 3);
  const int role = SPLIT ? ((wv >> 1) & 1) : (wv >> 2);
  const int half = wv >> 2;
  // row-wise parts: a wave owns the 16 columns `ct` (dconcat) / `kt` (dx) and the row tiles mine_*(rt)
  const int ct = SPLIT ? 2 * hp + (wv & 1) : (wv & 3);
  const int ktile = wv & 3;
  auto mine_dc = [&](int rt) { return SPLIT ? rt == (wv >> 1) : (rt & 1) == (wv >> 2); };
  auto mine_dx = [&](int rt) { return (rt & 1) == (wv >> 2); };
  T* Qs = reinterpret_cast<T*>(lds_bytes());   // [NR][P] q, later dq
  T* Ks = Qs + NR * P;     // k, later dk   (rows >= n_real: zero)
  T* Vs = Ks + NR * P;     // v, later dv   (rows >= n_real: zero)
  T* Ds = Vs + NR * P;     // dconcat
  T* Os = Ds + NR * P;     // out (per-head outputs, concatenated)
  T* Gt = Os + NR * P;     // g1 (BatchNorm-1 backward of dy; un-scaled)
  T* X0 = Gt + NR * P;     // x0 (raw: seen through bn0 on use)
  float* PE = reinterpret_cast<float*>(X0 + NR * P);   // [NR][PEP]
  float* ST = PE + NR * PEP;   // [H][NR][2]
  float* RS = ST + H * NR * 2; // [NR] degree scale of the rows
  float* DBO = RS + NR;        // [8 waves][64] fp32 column sums of (degree g1) over the rows each wave staged
  float* DBS = DBO + 8 * D;    // [8 waves][2][16] fp32 column sums of the wave's dq | dk, dv accumulators
  float* gv = DBS + 8 * 2 * 16;   // [5][64] scale, mean, rstd, m1, m2 of BatchNorm 1
  const T* gdy = reinterpret_cast<const T*>(a.dy);
  const T* gy1 = reinterpret_cast<const T*>(a.y1);
  const T* gqkv = reinterpret_cast<const T*>(a.qkv);
  const T* gout = reinterpret_cast<const T*>(a.out);
  const T* gd2 = reinterpret_cast<const T*>(a.dout2);
  const T* gpe = reinterpret_cast<const T*>(a.pe);
  const T* gx0 = reinterpret_cast<const T*>(a.x0);
  const bool gbn = a.bn1 != nullptr;
  // LayerNorm stack (feta_ln.h): dy is the gradient w.r.t. LN1(y1) and its LayerNorm backward is taken per row where the
  // gradient rows are staged; x0 holds pre-norm rows and is normalised where IT is staged
  const bool gln = a.ln1_gamma != nullptr, x0ln = a.x0_ln_gamma != nullptr;
  const bool has_y1 = a.y1 != nullptr;
  float* DGL = gv;                // [8 waves][2][64] column sums of (dy xhat1, dy) (gv itself is BatchNorm's)
  const bool has_pe = a.pe != nullptr;
  const bool want_sums = a.sum_out != nullptr;
  const bool xbn = a.bn0 != nullptr;
  const bool d2_f32 = a.dout2 != nullptr && a.dout2_f32 != 0 && sizeof(T) != sizeof(float);
  BB_STAMP(0);

  // ---- once per workgroup: BatchNorm-1 backward parameters, weight column slices ----------------------------------
  if (gbn) {
    float* scr = gv + 5 * D;
    const int cpre = min(tid, D - 1);
    const float bn_scale = a.bn1[cpre], bn_mean = a.bn1[2 * D + cpre], bn_rstd = a.bn1[3 * D + cpre];
    if (a.g_sum != nullptr) {
      reduce_partials_t<kBbThreads, 16>(a.g_sum, a.Gs, D, scr + 2 * D, scr);   // (all 512 threads: 256 rows in ONE batch)
      for (int c = tid; c < D; c += kBbThreads) {
        gv[3 * D + c] = scr[c] / (float)a.M;
        gv[4 * D + c] = scr[D + c] / (float)a.M;
        if (blockIdx.x == 0) {
          if (a.dbeta != nullptr) a.dbeta[c] = scr[c];
          if (a.dgamma != nullptr) a.dgamma[c] = scr[D + c];
          if (a.fin_out != nullptr) {
            a.fin_out[c] = gv[3 * D + c];
            a.fin_out[D + c] = gv[4 * D + c];
          }
        }
      }
    }
    if (tid < D) {
      gv[tid] = bn_scale;
      gv[D + tid] = bn_mean;
      gv[2 * D + tid] = bn_rstd;
    }
    __syncthreads();
  }
  float sc0 = 1.0f, sh0 = 0.0f, mean0[4] = {0.f, 0.f, 0.f, 0.f}, rstd0[4] = {0.f, 0.f, 0.f, 0.f};
  if (xbn) {
    sc0 = a.bn0[16 * ktile + lq];
    sh0 = a.bn0[D + 16 * ktile + lq];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mean0[r] = a.bn0[2 * D + 16 * ktile + 4 * g + r];
      rstd0[r] = a.bn0[3 * D + 16 * ktile + 4 * g + r];
    }
  }
  const int nm1 = a.N - 1;
  float sum1[4] = {0.f, 0.f, 0.f, 0.f}, sum2[4] = {0.f, 0.f, 0.f, 0.f};
  // one graph per workgroup (SPLIT: per two) up to kBbMaxGrid graphs; beyond that the workgroups walk the graphs
  // b, b + gridDim.x, ... and ADD each graph's weight-gradient tiles to their own partial row (read-modify-write by the
  // thread that wrote it: the row stays in the XCD's L2), the partial sums for the previous BatchNorm stay in registers
  const int nwg = main_grid;
  int b = (int)blockIdx.x - hp * a.B;
  const int lane0 = lane;
  do {
    // LOOP form: everything a lane derives from its id (tile addresses, predicates, offsets) is invariant in the graph
    // loop, and the compiler hoists all of it - hundreds of registers held across the attention phase (up to 676 B of
    // scratch per lane).  The lane id is laundered once per graph, so those values are recomputed where they are used.
    int lane_l = lane0;
    if (LOOP) FETA_OPAQUE_LANE(lane_l);
    const int lane = lane_l, tid = (wv << 6) | lane, lq = lane & 15, g = lane >> 4;
    const bool first = !LOOP || b < nwg;
    if (LOOP && !first) __syncthreads();   // the tiles of the previous graph have been consumed
    // The weight column slices below are loop-invariant loads: hoisted out of the graph loop they would be held across
    // the attention phase (112 registers per lane: the kernel spilled up to 1 KB per lane).  The pointers are
    // laundered once per graph so that the loads stay where they are used.
    const float* w_out_l = a.w_out;
    const float* w_in_l = a.w_in;
    if (LOOP) {
      FETA_OPAQUE_PTR(w_out_l);
      FETA_OPAQUE_PTR(w_in_l);
    }
    const feta_gcf w_out_p = (feta_gcf)w_out_l, w_in_p = (feta_gcf)w_in_l;
    const int n = a.n_real[b];
    BB_STAMP(1);
    auto grow = [&](int node) { return (int64_t)b * a.row_sb + (int64_t)min(node, nm1) * a.row_sn; };

    // ---- cooperative loads of the graph: NR * RV 16-byte vectors per 64-wide tensor --------------------------------
    constexpr int RI = (NR * RV + kBbThreads - 1) / kBbThreads;
    Vec qv[RI], kv[RI], vv[RI], ov[RI], dyv[RI], y1v[RI], x0v[RI], d2v[RI];
    float rsn[RI];
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int idx = min(tid + kBbThreads * i, NR * RV - 1), c4 = VEC * (idx % RV);
      const int64_t row = grow(idx / RV);
      qv[i] = L::ldv(gqkv + row * 3 * D + c4);
      kv[i] = L::ldv(gqkv + row * 3 * D + D + c4);
      vv[i] = L::ldv(gqkv + row * 3 * D + 2 * D + c4);
      ov[i] = L::ldv(gout + row * D + c4);
      dyv[i] = L::ldv(gdy + row * D + c4);
      // (no ternary on a whole vector: it is lowered to a private-memory select - the operands are read through a
      // pointer that falls back to a tensor which is always there)
      y1v[i] = L::ldv((has_y1 ? gy1 : gdy) + row * D + c4);
      x0v[i] = L::ldv(gx0 + row * D + c4);
      if (d2_f32) {   // the filter branch's gradient arrives as fp32 behind a bf16 stack: rounded here
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
          const float4 t4 = *reinterpret_cast<const float4*>(a.dout2 + row * D + c4 + e);
          f[e] = t4.x; f[e + 1] = t4.y; f[e + 2] = t4.z; f[e + 3] = t4.w;
        }
        d2v[i] = L::pack(f);
      } else {
        d2v[i] = L::ldv((a.dout2 != nullptr ? gd2 : gdy) + row * D + c4);
      }
      rsn[i] = a.rowscale != nullptr ? a.rowscale[row] : 1.0f;
    }
    constexpr int PEI = (NR * NR + kBbThreads - 1) / kBbThreads;
    float pev[PEI];
#pragma unroll
    for (int i = 0; i < PEI; ++i) {
      const int idx = tid + kBbThreads * i, qq = idx / NR, kk = idx - qq * NR;
      const float v = has_pe ? L::ld1(gpe + ((int64_t)b * a.N + min(qq, nm1)) * a.N + min(kk, nm1)) : 1.0f;
      pev[i] = (idx < NR * NR && qq < a.N && kk < a.N) ? v : 0.0f;
    }
    {
      const int hh = tid / (NR * 2), rem = tid - hh * NR * 2;   // H * NR * 2 <= 512
      const float sv = a.attn_stats[(((int64_t)b * H + min(hh, H - 1)) * a.N + min(rem >> 1, nm1)) * 2 + (rem & 1)];
      if (tid < H * NR * 2) ST[tid] = sv;
      if (tid < NR) RS[tid] = (a.rowscale != nullptr && tid < a.N) ? a.rowscale[grow(tid)] : (tid < a.N ? 1.0f : 0.0f);
    }
    // db_out = column sums of (degree g1): taken here from the fp32 values, BEFORE g1 is rounded into its tile (behind a
    // BatchNorm backward the true column sums of g1 are zero when degree = 1: a sum over a bf16 tile would be noise)
    float dbo[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) dbo[e] = 0.0f;
    // a thread stages the same VEC columns of every row it touches: its slices of gamma1 / gamma0 / beta0 and its column
    // sums of (dy xhat1, dy) are registers (requested here, through laundered pointers: they live through the staging
    // loop only, not across the graph loop)
    float gl1[VEC], gl0[VEC], bl0[VEC], dgam[VEC], dbet[VEC];
    {
      const float* p1 = gln ? a.ln1_gamma : a.w_out;
      const float* p0 = x0ln ? a.x0_ln_gamma : a.w_out;
      const float* q0 = x0ln ? a.x0_ln_beta : a.w_out;
      if (LOOP) {
        FETA_OPAQUE_PTR(p1);
        FETA_OPAQUE_PTR(p0);
        FETA_OPAQUE_PTR(q0);
      }
      const feta_gcf p1g = (feta_gcf)p1, p0g = (feta_gcf)p0, q0g = (feta_gcf)q0;
      const int c0v = VEC * (tid % RV);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        gl1[e] = p1g[c0v + e];
        gl0[e] = p0g[c0v + e];
        bl0[e] = q0g[c0v + e];
        dgam[e] = dbet[e] = 0.0f;
      }
    }
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int idx = tid + kBbThreads * i;
      if (idx < NR * RV) {
        const int node = idx / RV, c4 = VEC * (idx % RV), off = node * P + c4;
        const bool rk = node < a.N;       // rows beyond the padded length: zero everywhere
        const bool rkv = node < n;        // k / v rows of padded nodes: zero (the forward pass never wrote some of them)
        const bool dk2 = node < a.N && a.dout2 != nullptr;
        auto put = [&](T* dst, const Vec& v, bool keep) {
          float f[VEC];
          L::unpack(v, f);
#pragma unroll
          for (int e = 0; e < VEC; ++e) f[e] = keep ? f[e] : 0.0f;
          L::stv(dst + off, L::pack(f));
        };
        put(Qs, qv[i], rk);
        put(Ks, kv[i], rkv);
        put(Vs, vv[i], rkv);
        put(Os, ov[i], rk);
        if (x0ln) {   // x0 = LayerNorm(pre-norm row) * gamma0 + beta0 (a wave stages whole rows: wave-uniform)
          float f[VEC];
          L::unpack(x0v[i], f);
          ln_apply<VEC>(f, gl0, bl0, a.ln_eps);
#pragma unroll
          for (int e = 0; e < VEC; ++e) f[e] = rk ? f[e] : 0.0f;
          L::stv(X0 + off, L::pack(f));
        } else {
          put(X0, x0v[i], rk);
        }
        put(Ds, d2v[i], dk2);     // dout2; the product is added below
        float v[VEC];
        L::unpack(dyv[i], v);
        if (gln) {
          float yy[VEC];
          L::unpack(y1v[i], yy);
          ln_backward<VEC>(v, yy, gl1, a.ln_eps, rk, dgam, dbet);
        }
        if (gbn) {
          float yy[VEC];
          L::unpack(y1v[i], yy);
#pragma unroll
          for (int s = 0; s < VEC; ++s) {
            const int o = c4 + s;
            const float xh = (yy[s] - gv[D + o]) * gv[2 * D + o];
            v[s] = gv[o] * (v[s] - gv[3 * D + o] - xh * gv[4 * D + o]);
          }
        }
#pragma unroll
        for (int s = 0; s < VEC; ++s) {
          v[s] = rk ? v[s] : 0.0f;
          dbo[s] += rsn[i] * v[s];
        }
        L::stv(Gt + off, L::pack(v));
      }
    }
    // lanes l, l + RV, ... of a wave hold the same columns: one row per wave
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float sv = dbo[e];
      if (RV <= 8) sv += shfl_xor(sv, 8);
      sv += shfl_xor(sv, 16);
      sv += shfl_xor(sv, 32);
      if (lane < RV) DBO[wv * D + VEC * lane + e] = sv;
    }
    if (gln) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float sg = dgam[e], sb = dbet[e];
        if (RV <= 8) {
          sg += shfl_xor(sg, 8);
          sb += shfl_xor(sb, 8);
        }
        sg += shfl_xor(sg, 16);
        sb += shfl_xor(sb, 16);
        sg += shfl_xor(sg, 32);
        sb += shfl_xor(sb, 32);
        if (lane < RV) {
          DGL[(wv * 2 + 0) * D + VEC * lane + e] = sg;
          DGL[(wv * 2 + 1) * D + VEC * lane + e] = sb;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PEI; ++i) {
      const int idx = tid + kBbThreads * i;
      if (idx < NR * NR) PE[(idx / NR) * PEP + idx % NR] = pev[i];
    }
    __syncthreads();
    BB_STAMP(2);
    float* prow = a.partial + (int64_t)(SPLIT ? b : (int)blockIdx.x) *
                                  (a.partial_ld > 0 ? (int64_t)a.partial_ld
                                                    : (int64_t)(4 * D * D + 4 * D + (gln ? 2 * D : 0)));
    if (hp == 0 && tid < D) {
      float sv = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) sv += DBO[w8 * D + tid];
      acc_to(prow + D * D + tid, sv, first);
    }
    if (gln && hp == 0 && tid >= D && tid < 3 * D) {   // [dgamma1 | dbeta1] behind db_in
      const int which = (tid - D) / D, c = (tid - D) % D;
      float sv = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) sv += DGL[(w8 * 2 + which) * D + c];
      acc_to(prow + 4 * D * D + 4 * D + which * D + c, sv, first);
    }

    // ---- dconcat^T tiles (c = 16h + 4g + r, row = 16 rt + lq) = sum_o W_out[o][c] (degree g1)[row][o] (+ dout2) ----
    // (the wave's weight column slices are requested where they are used: held over the whole kernel they cost 64
    // registers that the attention phase needs - the kernel spilled)
    {
      Op woA[4];    // W_out[o = 16j+4g+s][c = 16 ct + lq]: dconcat columns of head ct
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const feta_gcf p = w_out_p + (int64_t)(16 * j + 4 * g) * D + 16 * ct + lq;
        woA[j] = L::mk(p[0], p[D], p[2 * D], p[3 * D]);
      }
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        if (!mine_dc(rt)) continue;
        const int rowl = 16 * rt + lq;
        RowOp<T, D> gf;
        load_row_op_scaled<T, D>(gf, Gt + rowl * P, g, RS[rowl]);
        f32x4 acc = zero4();
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = L::mma(woA[j], gf.o[j], acc);
        T* dst = Ds + rowl * P + 16 * ct + 4 * g;
        float d2[4];
        L::ld4(dst, d2);
        L::st4(dst, acc[0] + d2[0], acc[1] + d2[1], acc[2] + d2[2], acc[3] + d2[3]);
      }
    }
    __syncthreads();
    BB_STAMP(3);

    // ---- attention backward (attn_bwd_graph_kernel, attn.hip): role 0 dq over the head's query tiles, role 1 dk / dv
    const int co = DH * h;
    f32x4 r0[NT], r1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      r0[t] = zero4();
      r1[t] = zero4();
    }
    if (role == 0) {
      Op kf[NT], vf[NT], kb[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int rowl = 16 * t + lq;
        kf[t] = L::ld(Ks + rowl * P + co + 4 * g);
        vf[t] = L::ld(Vs + rowl * P + co + 4 * g);
        kb[t] = L::gather(Ks + (16 * t + 4 * g) * P + co + lq, P);
      }
#pragma unroll
      for (int qb = 0; qb < NT; ++qb) {
        const int q = 16 * qb + lq;
        const Op qf = L::ld_scaled(Qs + q * P + co + 4 * g, a.scale);   // (rows >= N of the tiles are zero)
        const Op dof = L::ld(Ds + q * P + co + 4 * g);
        const Op of = L::ld(Os + q * P + co + 4 * g);
        float delta = L::get(dof, 0) * L::get(of, 0) + L::get(dof, 1) * L::get(of, 1) + L::get(dof, 2) * L::get(of, 2) +
                      L::get(dof, 3) * L::get(of, 3);
        delta += shfl_xor(delta, 16);
        delta += shfl_xor(delta, 32);
        const float m = ST[(h * NR + q) * 2], z = ST[(h * NR + q) * 2 + 1];
        const float rinv = fast_rcp(fmaxf(z, 1e-6f));   // (v_rcp_f32, 1 ulp: a full-precision division is a dozen instructions)
        if (z < 1e-6f) delta = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          if (16 * kt >= n) continue;
          if (SPLIT && ((qb * NT + kt) & 1) != half) continue;   // the partner wave takes the other tile pairs
          const f32x4 s = L::mma(kf[kt], qf, zero4());
          const f32x4 da = L::mma(vf[kt], dof, zero4());
          float pd[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            const float p = key < n ? fast_exp(s[r] - m) * PE[q * PEP + key] * rinv : 0.0f;
            pd[r] = p * (da[r] - delta);
          }
          r0[qb] = L::mma(L::mk(pd[0], pd[1], pd[2], pd[3]), kb[kt], r0[qb]);  // (query 4g+r, c lq)
        }
      }
    } else {
      Op qf[NT], dof[NT], qb4[NT], dob[NT];
      float sd[NT][4], mq[NT][4], rz[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int rowl = 16 * t + lq;
        qf[t] = L::ld_scaled(Qs + rowl * P + co + 4 * g, a.scale);
        dof[t] = L::ld(Ds + rowl * P + co + 4 * g);
        const T* colq = Qs + (16 * t + 4 * g) * P + co + lq;
        const T* cold = Ds + (16 * t + 4 * g) * P + co + lq;
        const T* colo = Os + (16 * t + 4 * g) * P + co + lq;
        float qv4[4], dv4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = 16 * t + 4 * g + r;
          const bool ok = rr < a.N;
          dv4[r] = L::ld1(cold + r * P);
          qv4[r] = L::ld1(colq + r * P) * a.scale;
          sd[t][r] = row16_sum(dv4[r] * L::ld1(colo + r * P));   // delta[q = 4g + r]
          // softmax statistics of the query: once per query, not once per (query, key tile) - the division alone is
          // a dozen instructions and this role is the one the other waves wait for
          const float z = ST[(h * NR + rr) * 2 + 1];
          mq[t][r] = ST[(h * NR + rr) * 2];
          rz[t][r] = ok ? fast_rcp(fmaxf(z, 1e-6f)) : 0.0f;
          if (z < 1e-6f) sd[t][r] = 0.0f;
        }
        qb4[t] = L::mk(qv4[0], qv4[1], qv4[2], qv4[3]);
        dob[t] = L::mk(dv4[0], dv4[1], dv4[2], dv4[3]);
      }
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (16 * kt >= n) continue;
        const int key = 16 * kt + lq;
        const Op kf = L::ld(Ks + key * P + co + 4 * g);   // (rows >= n_real of the k / v tiles are zero)
        const Op vf = L::ld(Vs + key * P + co + 4 * g);
#pragma unroll
        for (int qb = 0; qb < NT; ++qb) {
          if (SPLIT && ((kt * NT + qb) & 1) != half) continue;
          const f32x4 s = L::mma(qf[qb], kf, zero4());
          const f32x4 da = L::mma(dof[qb], vf, zero4());
          float pp[4], ds[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = 16 * qb + 4 * g + r;
            pp[r] = key < n ? fast_exp(s[r] - mq[qb][r]) * PE[q * PEP + key] * rz[qb][r] : 0.0f;   // rz = 0: q >= N
            ds[r] = pp[r] * (da[r] - sd[qb][r]);
          }
          r1[kt] = L::mma(L::mk(pp[0], pp[1], pp[2], pp[3]), dob[qb], r1[kt]);    // dv (key 4g+r, c lq)
          r0[kt] = L::mma(L::mk(ds[0], ds[1], ds[2], ds[3]), qb4[qb], r0[kt]);    // dk
        }
      }
    }
    // fp32 column sums of this wave's dq | dk, dv accumulators (rows beyond the real / padded length are zero): the
    // bias gradient of in_proj, before the accumulators are rounded into the tiles
    {
      float c0 = 0.0f, c1 = 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          c0 += r0[t][r];
          c1 += r1[t][r];
        }
      c0 += shfl_xor(c0, 16);
      c0 += shfl_xor(c0, 32);
      c1 += shfl_xor(c1, 16);
      c1 += shfl_xor(c1, 32);
      if (g == 0) {
        DBS[(wv * 2 + 0) * 16 + lq] = role == 0 ? c0 * a.scale : c0;
        DBS[(wv * 2 + 1) * 16 + lq] = c1;
      }
    }
    // ---- dW_out = (degree g1)^T out: independent of the attention results, so the dq waves - which finish
    // well before the dk / dv waves (their set-up alone is twice as long) - take it while they would otherwise wait
    // at the barrier: o tile `ot` per wave, every column tile of this workgroup's heads
    // (contraction over the graph's rows: k-step s of group q4 is row 16 q4 + 4 s + g)
    constexpr int NWO = SPLIT ? 2 : 4;
    if (role == 0) {
      const int ot = SPLIT ? (wv & 1) + 2 * (wv >> 2) : (wv & 3);
      f32x4 aWo[NWO];
#pragma unroll
      for (int i = 0; i < NWO; ++i) aWo[i] = zero4();
#pragma unroll 1
      for (int q4 = 0; q4 < NT; ++q4) {
        const int rr = 16 * q4 + g;
        float gvv[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) gvv[s4] = RS[rr + 4 * s4] * L::ld1(Gt + (rr + 4 * s4) * P + 16 * ot + lq);
        const Op ga = L::mk(gvv[0], gvv[1], gvv[2], gvv[3]);   // (degree g1)[row][o = 16 ot + lq]
#pragma unroll
        for (int i = 0; i < NWO; ++i)
          aWo[i] = L::mma(ga, L::gather(Os + rr * P + 16 * (SPLIT ? 2 * hp + i : i) + lq, 4 * P), aWo[i]);
      }
#pragma unroll
      for (int i = 0; i < NWO; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
          acc_to(prow + (int64_t)(16 * ot + 4 * g + r) * D + 16 * (SPLIT ? 2 * hp + i : i) + lq, aWo[i][r], first);
        }
    }
    BB_STAMP(4);
    __syncthreads();   // every wave has taken its operands: the q / k / v tiles become dq / dk / dv
    // SPLIT: the two waves of a (head, role) hold partial sums over their tile pairs: one stores, then the other adds
#pragma unroll
    for (int pass = 0; pass < (SPLIT ? 2 : 1); ++pass) {
      if (!SPLIT || half == pass) {
        // (pass 0 overwrites)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * t + 4 * g + r;
            if (role == 0) {
              T* dq = Qs + rr * P + co + lq;
              L::st1(dq, (pass == 0 ? 0.0f : L::ld1(dq)) + r0[t][r] * a.scale);
            } else {
              T* dk = Ks + rr * P + co + lq;
              T* dv = Vs + rr * P + co + lq;
              L::st1(dk, (pass == 0 ? 0.0f : L::ld1(dk)) + r0[t][r]);
              L::st1(dv, (pass == 0 ? 0.0f : L::ld1(dv)) + r1[t][r]);
            }
          }
        }
      }
      __syncthreads();
    }

    BB_STAMP(5);
    // ---- dx^T tiles (k = 16 ktile + 4g + r, row) = sum_o W_in[o][k] dqkv[row][o] (+ g1[row][k]); sums for the previous
    // BatchNorm.  SPLIT: o runs over this pair's columns of dq | dk | dv only (the other columns of the tiles still hold
    // q | k | v of the other pair), the residual belongs to pair 0
    constexpr int NJX = SPLIT ? 2 : 4;
    f32x4 dxa[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) dxa[rt] = zero4();
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      Op wiA[NJX];   // W_in[o = 64 part + 16 jc + 4g+s][k = 16 ktile + lq], jc = this pair's (or every) head
#pragma unroll
      for (int j = 0; j < NJX; ++j) {
        const feta_gcf p = w_in_p + (int64_t)(64 * part + 16 * (SPLIT ? 2 * hp + j : j) + 4 * g) * D + 16 * ktile + lq;
        wiA[j] = L::mk(p[0], p[D], p[2 * D], p[3 * D]);
      }
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        if (!mine_dx(rt)) continue;
        const T* drow = (part == 0 ? Qs : (part == 1 ? Ks : Vs)) + (16 * rt + lq) * P + (SPLIT ? 32 * hp : 0);
#pragma unroll
        for (int j = 0; j < NJX; ++j) dxa[rt] = L::mma(wiA[j], L::ld(drow + 16 * j + 4 * g), dxa[rt]);
      }
    }
    T* dxo = reinterpret_cast<T*>((SPLIT && hp == 1) ? a.dx_b : a.dx);
    const float resw = (SPLIT && hp == 1) ? 0.0f : 1.0f;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      if (!mine_dx(rt)) continue;
      const int rowl = 16 * rt + lq;
      const f32x4 acc = dxa[rt];
      float res[4];
      L::ld4(Gt + rowl * P + 16 * ktile + 4 * g, res);
      const float v[4] = {acc[0] + resw * res[0], acc[1] + resw * res[1], acc[2] + resw * res[2], acc[3] + resw * res[3]};
      const bool rok = rowl < a.N;
      if (rok) L::st4(dxo + grow(rowl) * D + 16 * ktile + 4 * g, v[0], v[1], v[2], v[3]);
      if (want_sums) {
        float xx[4];
        L::ld4(X0 + rowl * P + 16 * ktile + 4 * g, xx);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = rok ? v[r] : 0.0f;
          sum1[r] += s1;
          sum2[r] += s1 * (xx[r] - mean0[r]) * rstd0[r];
        }
      }
    }
    if (want_sums) {
      // partial (sum dx, sum dx * xhat0) rows: one per (graph, row-tile parity); SPLIT: the two parities are added
      // through LDS first (pe is no longer needed), one row per workgroup = the same 2 B rows in both forms
      float s1v[4], s2v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1v[r] = row16_sum(sum1[r]);
        s2v[r] = row16_sum(sum2[r]);
      }
      if (SPLIT) {
        float* red = PE;   // [4 k tiles][2][16]
        if ((wv >> 2) == 1 && lq == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            red[(ktile * 2 + 0) * 16 + 4 * g + r] = s1v[r];
            red[(ktile * 2 + 1) * 16 + 4 * g + r] = s2v[r];
          }
        }
        __syncthreads();
        if ((wv >> 2) == 0 && lq == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            a.sum_out[((int64_t)blockIdx.x * 2 + 0) * D + 16 * ktile + 4 * g + r] = s1v[r] + red[(ktile * 2 + 0) * 16 + 4 * g + r];
            a.sum_out[((int64_t)blockIdx.x * 2 + 1) * D + 16 * ktile + 4 * g + r] = s2v[r] + red[(ktile * 2 + 1) * 16 + 4 * g + r];
          }
        }
      } else if (lq == 0 && (!LOOP || b + nwg >= a.B)) {   // (the last graph of this workgroup: sum1 / sum2 ran over all)
        const int64_t prow2 = (int64_t)blockIdx.x * 2 + (wv >> 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a.sum_out[(prow2 * 2 + 0) * D + 16 * ktile + 4 * g + r] = s1v[r];
          a.sum_out[(prow2 * 2 + 1) * D + 16 * ktile + 4 * g + r] = s2v[r];
        }
      }
    }

    BB_STAMP(6);
    // ---- dW_in of the graph, contraction over its rows (rows >= N are zero in every tile): dW_in[o][k = 16 ktile + lq],
    // NWI row tiles per wave (row tile = q | k | v part x head); SPLIT: only this pair's rows
    constexpr int NWI = SPLIT ? 3 : 6;
    const int grp = wv >> 2;
    f32x4 aWi[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) aWi[i] = zero4();
    auto wi_part = [&](int i) { return SPLIT ? (3 * grp + i) >> 1 : (6 * grp + i) >> 2; };
    auto wi_head = [&](int i) { return SPLIT ? 2 * hp + ((3 * grp + i) & 1) : ((6 * grp + i) & 3); };
#pragma unroll 1
    for (int q4 = 0; q4 < NT; ++q4) {
      const int rr = 16 * q4 + g;
      float xv4[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) xv4[s4] = L::ld1(X0 + (rr + 4 * s4) * P + 16 * ktile + lq) * sc0 + sh0;
      const Op xb = L::mk(xv4[0], xv4[1], xv4[2], xv4[3]);    // x0 through its BatchNorm, [row][k = 16 ktile + lq]
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int part = wi_part(i);
        const T* src = part == 0 ? Qs : (part == 1 ? Ks : Vs);
        aWi[i] = L::mma(L::gather(src + rr * P + 16 * wi_head(i) + lq, 4 * P), xb, aWi[i]);
      }
    }
    // ---- partial row of this graph: [dW_out (64 x 64) | db_out (64) | dW_in (192 x 64) | db_in (192)] --------------
    float* pWi = prow + D * D + D;
    float* pbi = pWi + 3 * D * D;
#pragma unroll
    for (int i = 0; i < NWI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
      {
        acc_to(pWi + (int64_t)(64 * wi_part(i) + 16 * wi_head(i) + 4 * g + r) * D + 16 * ktile + lq, aWi[i][r], first);
      }
    // db_in from the fp32 column sums the attention waves left (DBS: written before two barriers ago)
    if (tid < 3 * D) {
      const int part = tid >> 6, hh = (tid >> 4) & 3, c = tid & 15;
      const int rl = part == 0 ? 0 : 1, which = part == 2 ? 1 : 0;
      if (SPLIT) {
        if ((hh >> 1) == hp) {
          const int wa = (hh & 1) + 2 * rl;
          pbi[tid] = DBS[(wa * 2 + which) * 16 + c] + DBS[((wa + 4) * 2 + which) * 16 + c];   // (one graph per pair)
        }
      } else {
        acc_to(pbi + tid, DBS[((hh + 4 * rl) * 2 + which) * 16 + c], first);
      }
    }
    b += nwg;
  } while (LOOP && b < a.B);
  BB_STAMP(7);
  FETA_RT_LAUNCH_DONE(feta_bbwd_launch);
}

template <class T, int NT>
int launch_block_bwd(const BwdArgs& a, const feta_colsum_seg* segs, int nseg, hipStream_t stream) {
  size_t lds = block_bwd_lds_bytes<T>(NT, a.bn1 != nullptr, a.ln1_gamma != nullptr);
  const int grid = feta_attn_block_bwd_blocks(a.B);
  ColsumPlan plan{};
  const int tiles = plan_colsum(segs, nseg, plan);
  if (tiles > 0 && lds < sizeof(float) * colsum_role_lds_floats(kBbThreads)) lds = sizeof(float) * colsum_role_lds_floats(kBbThreads);
  if (a.dx_b != nullptr) {   // two workgroups per graph
    auto kern = attn_block_bwd_kernel<T, NT, true, false>;
    static LdsSeen lds_seen;
    allow_dynamic_lds(kern, lds, lds_seen);
    hipLaunchKernelGGL(kern, dim3(2 * a.B + tiles), dim3(kBbThreads), lds, stream, a, plan, 2 * a.B);
  } else if (grid == a.B) {
    auto kern = attn_block_bwd_kernel<T, NT, false, false>;
    static LdsSeen lds_seen;
    allow_dynamic_lds(kern, lds, lds_seen);
    hipLaunchKernelGGL(kern, dim3(grid + tiles), dim3(kBbThreads), lds, stream, a, plan, grid);
  } else {
    auto kern = attn_block_bwd_kernel<T, NT, false, true>;
    static LdsSeen lds_seen;
    allow_dynamic_lds(kern, lds, lds_seen);
    hipLaunchKernelGGL(kern, dim3(grid + tiles), dim3(kBbThreads), lds, stream, a, plan, grid);
  }
  return check_launch("feta_attn_block_bwd");
}

template <class T>
int dispatch_block_bwd(const BwdArgs& a, const feta_colsum_seg* segs, int nseg, hipStream_t stream) {
  switch ((a.N + 15) / 16) {
    case 1: return launch_block_bwd<T, 1>(a, segs, nseg, stream);
    case 2: return launch_block_bwd<T, 2>(a, segs, nseg, stream);
    case 3: return launch_block_bwd<T, 3>(a, segs, nseg, stream);
    default: return launch_block_bwd<T, 4>(a, segs, nseg, stream);
  }
}

}  // namespace feta

using namespace feta;

#ifdef FETA_TIMING
extern "C" int feta_debug_bbwd_stamps(unsigned long long* out256) {
  return (int)hipMemcpyFromSymbol(out256, HIP_SYMBOL(feta_bbwd_stamps), sizeof(unsigned long long) * 256);
}
#endif

extern "C" int feta_attn_block_bwd_supported(int N, int d_model, int heads) {
  return (d_model == kBbD && heads == kBbH && N >= 1 && N <= 64) ? 1 : 0;
}

/* partial rows of a launch = its workgroups: one per graph up to kBbMaxGrid (FETA_BLOCK_BWD_MAX_GRID: tests force the
 * loop), beyond that the workgroups walk several graphs */
extern "C" int feta_attn_block_bwd_blocks(int B) {
  int cap = kBbMaxGrid;
  if (const char* e = getenv("FETA_BLOCK_BWD_MAX_GRID")) cap = atoi(e) > 0 ? atoi(e) : cap;
  return B < 1 ? 0 : (B < cap ? B : cap);
}

extern "C" int feta_attn_block_bwd(const feta_attn_block_grad* d, feta_stream_t stream) {
  return feta_attn_block_bwd_sums(d, nullptr, 0, stream);
}

extern "C" int feta_attn_block_bwd_sums(const feta_attn_block_grad* d, const feta_colsum_seg* segs, int nseg,
                                        feta_stream_t stream) {
  FETA_REQUIRE(d != nullptr, "attn_block_bwd: null descriptor");
  FETA_REQUIRE(nseg >= 0 && nseg <= FETA_COLSUM_MAX_SEGS && (nseg == 0 || segs != nullptr),
               "attn_block_bwd: 0..%d column-sum segments", FETA_COLSUM_MAX_SEGS);
  for (int i = 0; i < nseg; ++i) FETA_REQUIRE(colsum_seg_ok(segs[i]), "attn_block_bwd: bad segment %d", i);
  const BwdArgs& a = *d;
  FETA_REQUIRE(a.dy && a.w_out && a.w_in && a.qkv && a.out && a.n_real && a.attn_stats && a.x0 && a.dx && a.partial,
               "attn_block_bwd: null pointer");
  FETA_REQUIRE(a.B > 0, "attn_block_bwd: B=%d", a.B);
  FETA_REQUIRE(a.dx_b == nullptr || feta_attn_block_bwd_blocks(a.B) == a.B,
               "attn_block_bwd: the two-workgroup form (dx_b) needs one workgroup per graph (B=%d > %d)", a.B,
               feta_attn_block_bwd_blocks(a.B));
  FETA_REQUIRE(a.N >= 1 && a.N <= 64 && a.M == a.B * a.N, "attn_block_bwd: N=%d outside [1,64] or M != B*N", a.N);
  FETA_REQUIRE(!a.y1 || a.ln1_gamma || (a.bn1 && a.g_sum && a.Gs > 0), "attn_block_bwd: y1 needs bn1, g_sum, Gs - or ln1_gamma");
  FETA_REQUIRE(!a.bn1 || a.y1, "attn_block_bwd: bn1 needs y1");
  FETA_REQUIRE(!a.ln1_gamma || (a.y1 && !a.bn1 && !a.g_sum), "attn_block_bwd: ln1_gamma needs y1 and excludes bn1 / g_sum");
  FETA_REQUIRE(!a.x0_ln_gamma || (a.x0_ln_beta && !a.bn0 && !a.sum_out), "attn_block_bwd: x0_ln_gamma needs x0_ln_beta and excludes bn0 / sum_out");
  FETA_REQUIRE(!a.sum_out || a.bn0, "attn_block_bwd: sum_out needs bn0");
  FETA_REQUIRE(aligned16(a.dy) && aligned16(a.qkv) && aligned16(a.out) && aligned16(a.x0) && aligned16(a.dx) &&
               aligned16(a.y1) && aligned16(a.dout2) && aligned16(a.g_sum) && aligned16(a.dx_b),
               "attn_block_bwd: tensors must be 16-byte aligned");
  FETA_REQUIRE(a.dtype == FETA_F32 || a.dtype == FETA_BF16, "attn_block_bwd: dtype %d", a.dtype);
  if (a.dtype == FETA_BF16) return dispatch_block_bwd<bf16_t>(a, segs, nseg, (hipStream_t)stream);
  return dispatch_block_bwd<float>(a, segs, nseg, (hipStream_t)stream);
}
