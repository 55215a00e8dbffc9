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
#include "feta_rowops.h"

namespace feta {

typedef feta_attn_block_grad BwdArgs;  // include/feta_hip.h

constexpr int kBbD = 64, kBbH = 4, kBbDH = 16;
constexpr int kBbP = kBbD + 4;   // pitch of a staged 64-float row
constexpr int kBbThreads = 512;
constexpr int kBbMaxGrid = 256;

__host__ __device__ inline int block_bwd_lds_floats(int nt, bool gbn) {
  const int nr = 16 * nt;
  return 7 * nr * kBbP          // q, k, v (later dq, dk, dv), dconcat, out, g1, x0
         + nr * (nr + 1)        // pe
         + kBbH * nr * 2 + nr   // softmax statistics, row scale
         + (gbn ? 5 * kBbD + reduce_scratch_floats(kBbD) : 0);
}

#ifdef FETA_TIMING
__device__ unsigned long long feta_bbwd_stamps[4 * 8 * 8];   // FETA_RT_STAMP (feta_rowops.h), tools/block_timing.py
__device__ unsigned int feta_bbwd_launch;
#endif
#define BB_STAMP(i) FETA_RT_STAMP(feta_bbwd_stamps, feta_bbwd_launch, i)

template <int NT, bool SPLIT>
__global__ __launch_bounds__(kBbThreads) void attn_block_bwd_kernel(BwdArgs a) {
  constexpr int D = kBbD, DH = kBbDH, H = kBbH, P = kBbP, NR = 16 * NT, PEP = NR + 1;
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
  float* Qs = feta_lds;        // [NR][P] q, later dq
  float* Ks = Qs + NR * P;     // k, later dk
  float* Vs = Ks + NR * P;     // v, later dv
  float* Ds = Vs + NR * P;     // dconcat
  float* Os = Ds + NR * P;     // out (per-head outputs, concatenated)
  float* Gt = Os + NR * P;     // g1 (BatchNorm-1 backward of dy; un-scaled)
  float* X0 = Gt + NR * P;     // x0 (raw: seen through bn0 on use)
  float* PE = X0 + NR * P;     // [NR][PEP]
  float* ST = PE + NR * PEP;   // [H][NR][2]
  float* RS = ST + H * NR * 2; // [NR] degree scale of the rows
  float* gv = RS + NR;         // [5][64] scale, mean, rstd, m1, m2 of BatchNorm 1
  const bool gbn = a.y1 != nullptr;
  const bool has_pe = a.pe != nullptr;
  const bool want_sums = a.sum_out != nullptr;
  const bool xbn = a.bn0 != nullptr;
  BB_STAMP(0);

  // ---- once per workgroup: BatchNorm-1 backward parameters, weight column slices ----------------------------------
  if (gbn) {
    float* scr = gv + 5 * D;
    const int cpre = min(tid, D - 1);
    const float bn_scale = a.bn1[cpre], bn_mean = a.bn1[2 * D + cpre], bn_rstd = a.bn1[3 * D + cpre];
    if (a.g_sum != nullptr) {
      reduce_partials(a.g_sum, a.Gs, D, scr + 2 * D, scr);
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
  {
    const int b = (int)blockIdx.x - hp * a.B;   // one graph per workgroup (SPLIT: per two)
    const int n = a.n_real[b];
    BB_STAMP(1);
    float sum1[4] = {0.f, 0.f, 0.f, 0.f}, sum2[4] = {0.f, 0.f, 0.f, 0.f};
    auto grow = [&](int node) { return (int64_t)b * a.row_sb + (int64_t)min(node, nm1) * a.row_sn; };

    // ---- cooperative loads of the graph: NR * 16 float4 per 64-wide tensor ------------------------------------------
    constexpr int RI = (NR * 16 + kBbThreads - 1) / kBbThreads;
    float4 qv[RI], kv[RI], vv[RI], ov[RI], dyv[RI], y1v[RI], x0v[RI], d2v[RI];
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int idx = min(tid + kBbThreads * i, NR * 16 - 1), c4 = 4 * (idx & 15);
      const int64_t row = grow(idx >> 4);
      qv[i] = *reinterpret_cast<const float4*>(a.qkv + row * 3 * D + c4);
      kv[i] = *reinterpret_cast<const float4*>(a.qkv + row * 3 * D + D + c4);
      vv[i] = *reinterpret_cast<const float4*>(a.qkv + row * 3 * D + 2 * D + c4);
      ov[i] = *reinterpret_cast<const float4*>(a.out + row * D + c4);
      dyv[i] = *reinterpret_cast<const float4*>(a.dy + row * D + c4);
      // (no ternary on a whole float4: it is lowered to a private-memory select - the operands are read through a
      // pointer that falls back to a tensor which is always there)
      y1v[i] = *reinterpret_cast<const float4*>((gbn ? a.y1 : a.dy) + row * D + c4);
      x0v[i] = *reinterpret_cast<const float4*>(a.x0 + row * D + c4);
      d2v[i] = *reinterpret_cast<const float4*>((a.dout2 != nullptr ? a.dout2 : a.dy) + row * D + c4);
    }
    constexpr int PEI = (NR * NR + kBbThreads - 1) / kBbThreads;
    float pev[PEI];
#pragma unroll
    for (int i = 0; i < PEI; ++i) {
      const int idx = tid + kBbThreads * i, qq = idx / NR, kk = idx - qq * NR;
      const float v = has_pe ? a.pe[((int64_t)b * a.N + min(qq, nm1)) * a.N + min(kk, nm1)] : 1.0f;
      pev[i] = (idx < NR * NR && qq < a.N && kk < a.N) ? v : 0.0f;
    }
    {
      const int hh = tid / (NR * 2), rem = tid - hh * NR * 2;   // H * NR * 2 <= 512
      const float sv = a.attn_stats[(((int64_t)b * H + min(hh, H - 1)) * a.N + min(rem >> 1, nm1)) * 2 + (rem & 1)];
      if (tid < H * NR * 2) ST[tid] = sv;
      if (tid < NR) RS[tid] = (a.rowscale != nullptr && tid < a.N) ? a.rowscale[grow(tid)] : (tid < a.N ? 1.0f : 0.0f);
    }
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int idx = tid + kBbThreads * i;
      if (idx < NR * 16) {
        const int node = idx >> 4, c4 = 4 * (idx & 15), off = node * P + c4;
        const float rk = node < a.N ? 1.0f : 0.0f;      // rows beyond the padded length: zero everywhere
        const float dk2 = (node < a.N && a.dout2 != nullptr) ? 1.0f : 0.0f;
        auto put = [&](float* dst, const float4& v, float m) {
          *reinterpret_cast<float4*>(dst + off) = make_float4(m * v.x, m * v.y, m * v.z, m * v.w);
        };
        put(Qs, qv[i], rk);
        put(Ks, kv[i], rk);
        put(Vs, vv[i], rk);
        put(Os, ov[i], rk);
        put(X0, x0v[i], rk);
        put(Ds, d2v[i], dk2);     // dout2; the product is added below
        float v[4] = {dyv[i].x, dyv[i].y, dyv[i].z, dyv[i].w};
        if (gbn) {
          const float yy[4] = {y1v[i].x, y1v[i].y, y1v[i].z, y1v[i].w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int o = c4 + s;
            const float xh = (yy[s] - gv[D + o]) * gv[2 * D + o];
            v[s] = gv[o] * (v[s] - gv[3 * D + o] - xh * gv[4 * D + o]);
          }
        }
        *reinterpret_cast<float4*>(Gt + off) = make_float4(rk * v[0], rk * v[1], rk * v[2], rk * v[3]);
      }
    }
#pragma unroll
    for (int i = 0; i < PEI; ++i) {
      const int idx = tid + kBbThreads * i;
      if (idx < NR * NR) PE[(idx / NR) * PEP + idx % NR] = pev[i];
    }
    __syncthreads();
    BB_STAMP(2);

    // ---- dconcat^T tiles (c = 16h + 4g + r, row = 16 rt + lq) = sum_o W_out[o][c] (degree g1)[row][o] (+ dout2) ----
    // (the wave's weight column slices are requested where they are used: held over the whole kernel they cost 64
    // registers that the attention phase needs - the kernel spilled)
    float woA[4][4];    // W_out[o = 16j+4g+s][c = 16 ct + lq]: dconcat columns of head ct
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) woA[j][s] = a.w_out[(int64_t)(16 * j + 4 * g + s) * D + 16 * ct + lq];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      if (!mine_dc(rt)) continue;
      const int rowl = 16 * rt + lq;
      Feat<D> gf;
      load_row<D>(gf, Gt + rowl * P, g, RS[rowl]);
      f32x4 acc = zero4();
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma16(woA[j][s], gf.f[j][s], acc);
      float4* dst = reinterpret_cast<float4*>(Ds + rowl * P + 16 * ct + 4 * g);
      const float4 d2 = *dst;
      *dst = make_float4(acc[0] + d2.x, acc[1] + d2.y, acc[2] + d2.z, acc[3] + d2.w);
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
      Feat<DH> kf[NT], vf[NT];
      float kb[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int rowl = 16 * t + lq;
        load_row<DH>(kf[t], Ks + rowl * P + co, g);
        load_row<DH>(vf[t], Vs + rowl * P + co, g);
        if (rowl >= n) {
#pragma unroll
          for (int s = 0; s < 4; ++s) kf[t].f[0][s] = vf[t].f[0][s] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = 16 * t + 4 * g + r;
          kb[t][r] = rr < n ? Ks[rr * P + co + lq] : 0.0f;
        }
      }
#pragma unroll
      for (int qb = 0; qb < NT; ++qb) {
        const int q = 16 * qb + lq;
        const bool qok = q < a.N;
        Feat<DH> qf, dof, of;
        load_row<DH>(qf, Qs + q * P + co, g, a.scale);
        load_row<DH>(dof, Ds + q * P + co, g);
        load_row<DH>(of, Os + q * P + co, g);
        if (!qok) {
#pragma unroll
          for (int s = 0; s < 4; ++s) qf.f[0][s] = dof.f[0][s] = 0.0f;
        }
        float delta = dof.f[0][0] * of.f[0][0] + dof.f[0][1] * of.f[0][1] + dof.f[0][2] * of.f[0][2] +
                      dof.f[0][3] * of.f[0][3];
        delta += shfl_xor(delta, 16);
        delta += shfl_xor(delta, 32);
        const float m = ST[(h * NR + q) * 2], z = ST[(h * NR + q) * 2 + 1];
        const float rinv = 1.0f / fmaxf(z, 1e-6f);
        if (z < 1e-6f) delta = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          if (16 * kt >= n) continue;
          if (SPLIT && ((qb * NT + kt) & 1) != half) continue;   // the partner wave takes the other tile pairs
          const f32x4 s = dot_rows<DH>(kf[kt], qf, zero4());
          const f32x4 da = dot_rows<DH>(vf[kt], dof, zero4());
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            const float p = key < n ? fast_exp(s[r] - m) * PE[q * PEP + key] * rinv : 0.0f;
            r0[qb] = mfma16(p * (da[r] - delta), kb[kt][r], r0[qb]);  // (query 4g+r, c lq)
          }
        }
      }
    } else {
      Feat<DH> qf[NT], dof[NT];
      float qb4[NT][4], dob[NT][4], sd[NT][4], mq[NT][4], rz[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int rowl = 16 * t + lq;
        load_row<DH>(qf[t], Qs + rowl * P + co, g, a.scale);
        load_row<DH>(dof[t], Ds + rowl * P + co, g);
        if (rowl >= a.N) {
#pragma unroll
          for (int s = 0; s < 4; ++s) qf[t].f[0][s] = dof[t].f[0][s] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = 16 * t + 4 * g + r;
          const bool ok = rr < a.N;
          const float dvv = Ds[rr * P + co + lq];
          qb4[t][r] = ok ? Qs[rr * P + co + lq] * a.scale : 0.0f;
          dob[t][r] = ok ? dvv : 0.0f;
          sd[t][r] = row16_sum(ok ? dvv * Os[rr * P + co + lq] : 0.0f);   // delta[q = 4g + r]
          // softmax statistics of the query: once per query, not once per (query, key tile) - the division alone is
          // a dozen instructions and this role is the one the other waves wait for
          const float z = ST[(h * NR + rr) * 2 + 1];
          mq[t][r] = ST[(h * NR + rr) * 2];
          rz[t][r] = ok ? 1.0f / fmaxf(z, 1e-6f) : 0.0f;
          if (z < 1e-6f) sd[t][r] = 0.0f;
        }
      }
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (16 * kt >= n) continue;
        const int key = 16 * kt + lq;
        Feat<DH> kf, vf;
        load_row<DH>(kf, Ks + key * P + co, g);
        load_row<DH>(vf, Vs + key * P + co, g);
        if (key >= n) {
#pragma unroll
          for (int s = 0; s < 4; ++s) kf.f[0][s] = vf.f[0][s] = 0.0f;
        }
#pragma unroll
        for (int qb = 0; qb < NT; ++qb) {
          if (SPLIT && ((kt * NT + qb) & 1) != half) continue;
          const f32x4 s = dot_rows<DH>(qf[qb], kf, zero4());
          const f32x4 da = dot_rows<DH>(dof[qb], vf, zero4());
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = 16 * qb + 4 * g + r;
            const float p = key < n ? fast_exp(s[r] - mq[qb][r]) * PE[q * PEP + key] * rz[qb][r] : 0.0f;   // rz = 0: q >= N
            const float ds = p * (da[r] - sd[qb][r]);
            r1[kt] = mfma16(p, dob[qb][r], r1[kt]);    // dv (key 4g+r, c lq)
            r0[kt] = mfma16(ds, qb4[qb][r], r0[kt]);   // dk
          }
        }
      }
    }
    // ---- dW_out = (degree g1)^T out and db_out: independent of the attention results, so the dq waves - which finish
    // well before the dk / dv waves (their set-up alone is twice as long) - take them while they would otherwise wait
    // at the barrier: o tile `ot` per wave, every column tile of this workgroup's heads
    constexpr int NWO = SPLIT ? 2 : 4;
    float* prow = a.partial + (int64_t)b * (a.partial_ld > 0 ? (int64_t)a.partial_ld : (int64_t)(4 * D * D + 4 * D));
    if (role == 0) {
      const int ot = SPLIT ? (wv & 1) + 2 * (wv >> 2) : (wv & 3);
      f32x4 aWo[NWO];
#pragma unroll
      for (int i = 0; i < NWO; ++i) aWo[i] = zero4();
      float dbo = 0.0f;
#pragma unroll 2
      for (int st = 0; st < NR / 4; ++st) {
        const int rr = 4 * st + g;
        const float ga = RS[rr] * Gt[rr * P + 16 * ot + lq];   // (degree g1)[row][o = 16 ot + lq]
        dbo += ga;
#pragma unroll
        for (int i = 0; i < NWO; ++i) aWo[i] = mfma16(ga, Os[rr * P + 16 * (SPLIT ? 2 * hp + i : i) + lq], aWo[i]);
      }
#pragma unroll
      for (int i = 0; i < NWO; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          prow[(int64_t)(16 * ot + 4 * g + r) * D + 16 * (SPLIT ? 2 * hp + i : i) + lq] = aWo[i][r];
      dbo += shfl_xor(dbo, 16);
      dbo += shfl_xor(dbo, 32);
      if (hp == 0 && g == 0) prow[D * D + 16 * ot + lq] = dbo;
    }
    BB_STAMP(4);
    __syncthreads();   // every wave has taken its operands: the q / k / v tiles become dq / dk / dv
    // SPLIT: the two waves of a (head, role) hold partial sums over their tile pairs: one stores, then the other adds
#pragma unroll
    for (int pass = 0; pass < (SPLIT ? 2 : 1); ++pass) {
      if (!SPLIT || half == pass) {
        // (pass 0 overwrites: k / v rows of padded nodes were never written by the forward pass)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * t + 4 * g + r;
            if (role == 0) {
              float* dq = Qs + rr * P + co + lq;
              *dq = (pass == 0 ? 0.0f : *dq) + r0[t][r] * a.scale;
            } else {
              float* dk = Ks + rr * P + co + lq;
              float* dv = Vs + rr * P + co + lq;
              *dk = (pass == 0 ? 0.0f : *dk) + r0[t][r];
              *dv = (pass == 0 ? 0.0f : *dv) + r1[t][r];
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
      float wiA[NJX][4];   // W_in[o = 64 part + 16 jc + 4g+s][k = 16 ktile + lq], jc = this pair's (or every) head
#pragma unroll
      for (int j = 0; j < NJX; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          wiA[j][s] = a.w_in[(int64_t)(64 * part + 16 * (SPLIT ? 2 * hp + j : j) + 4 * g + s) * D + 16 * ktile + lq];
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        if (!mine_dx(rt)) continue;
        const float* drow = (part == 0 ? Qs : (part == 1 ? Ks : Vs)) + (16 * rt + lq) * P + (SPLIT ? 32 * hp : 0);
#pragma unroll
        for (int j = 0; j < NJX; ++j) {
          const float4 dq4 = *reinterpret_cast<const float4*>(drow + 16 * j + 4 * g);
          dxa[rt] = mfma16(wiA[j][0], dq4.x, dxa[rt]);
          dxa[rt] = mfma16(wiA[j][1], dq4.y, dxa[rt]);
          dxa[rt] = mfma16(wiA[j][2], dq4.z, dxa[rt]);
          dxa[rt] = mfma16(wiA[j][3], dq4.w, dxa[rt]);
        }
      }
    }
    float* dxo = (SPLIT && hp == 1) ? a.dx_b : a.dx;
    const float resw = (SPLIT && hp == 1) ? 0.0f : 1.0f;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      if (!mine_dx(rt)) continue;
      const int rowl = 16 * rt + lq;
      const f32x4 acc = dxa[rt];
      const float4 res = *reinterpret_cast<const float4*>(Gt + rowl * P + 16 * ktile + 4 * g);
      const float v[4] = {acc[0] + resw * res.x, acc[1] + resw * res.y, acc[2] + resw * res.z, acc[3] + resw * res.w};
      const bool rok = rowl < a.N;
      if (rok) *reinterpret_cast<float4*>(dxo + grow(rowl) * D + 16 * ktile + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
      if (want_sums) {
        const float4 xr = *reinterpret_cast<const float4*>(X0 + rowl * P + 16 * ktile + 4 * g);
        const float xx[4] = {xr.x, xr.y, xr.z, xr.w};
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
      } else if (lq == 0) {
        const int64_t prow = (int64_t)blockIdx.x * 2 + (wv >> 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a.sum_out[(prow * 2 + 0) * D + 16 * ktile + 4 * g + r] = s1v[r];
          a.sum_out[(prow * 2 + 1) * D + 16 * ktile + 4 * g + r] = s2v[r];
        }
      }
    }

    BB_STAMP(6);
    // ---- dW_in of the graph, contraction over its rows (rows >= N are zero in every tile): dW_in[o][k = 16 ktile + lq],
    // NWI row tiles per wave (row tile = q | k | v part x head); SPLIT: only this pair's rows
    constexpr int NWI = SPLIT ? 3 : 6;
    const int grp = wv >> 2;
    f32x4 aWi[NWI];
    float dbi[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      aWi[i] = zero4();
      dbi[i] = 0.0f;
    }
    auto wi_part = [&](int i) { return SPLIT ? (3 * grp + i) >> 1 : (6 * grp + i) >> 2; };
    auto wi_head = [&](int i) { return SPLIT ? 2 * hp + ((3 * grp + i) & 1) : ((6 * grp + i) & 3); };
    for (int st = 0; st < NR / 4; ++st) {
      const int rr = 4 * st + g;
      const float xb = X0[rr * P + 16 * ktile + lq] * sc0 + sh0;    // x0 through its BatchNorm, [row][k = 16 ktile + lq]
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int part = wi_part(i);
        const float* src = part == 0 ? Qs : (part == 1 ? Ks : Vs);
        const float da = src[rr * P + 16 * wi_head(i) + lq];
        dbi[i] += da;
        aWi[i] = mfma16(da, xb, aWi[i]);
      }
    }
    // ---- partial row of this graph: [dW_out (64 x 64) | db_out (64) | dW_in (192 x 64) | db_in (192)] --------------
    float* pWi = prow + D * D + D;
    float* pbi = pWi + 3 * D * D;
#pragma unroll
    for (int i = 0; i < NWI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pWi[(int64_t)(64 * wi_part(i) + 16 * wi_head(i) + 4 * g + r) * D + 16 * ktile + lq] = aWi[i][r];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      float s = dbi[i];
      s += shfl_xor(s, 16);
      s += shfl_xor(s, 32);
      if (ktile == 0 && g == 0) pbi[64 * wi_part(i) + 16 * wi_head(i) + lq] = s;
    }
  }
  BB_STAMP(7);
  FETA_RT_LAUNCH_DONE(feta_bbwd_launch);
}

template <int NT>
int launch_block_bwd(const BwdArgs& a, hipStream_t stream) {
  const size_t lds = sizeof(float) * block_bwd_lds_floats(NT, a.y1 != nullptr);
  if (a.dx_b != nullptr) {   // two workgroups per graph
    auto kern = attn_block_bwd_kernel<NT, true>;
    static LdsSeen lds_seen;
    allow_dynamic_lds(kern, lds, lds_seen);
    hipLaunchKernelGGL(kern, dim3(2 * a.B), dim3(kBbThreads), lds, stream, a);
  } else {
    auto kern = attn_block_bwd_kernel<NT, false>;
    static LdsSeen lds_seen;
    allow_dynamic_lds(kern, lds, lds_seen);
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(kBbThreads), lds, stream, a);
  }
  return check_launch("feta_attn_block_bwd");
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

/* partial rows of a launch (= graphs), 0 if the batch is beyond the one-workgroup-per-graph form */
extern "C" int feta_attn_block_bwd_blocks(int B) { return (B >= 1 && B <= kBbMaxGrid) ? B : 0; }

extern "C" int feta_attn_block_bwd(const feta_attn_block_grad* d, feta_stream_t stream) {
  FETA_REQUIRE(d != nullptr, "attn_block_bwd: null descriptor");
  const BwdArgs& a = *d;
  FETA_REQUIRE(a.dy && a.w_out && a.w_in && a.qkv && a.out && a.n_real && a.attn_stats && a.x0 && a.dx && a.partial,
               "attn_block_bwd: null pointer");
  FETA_REQUIRE(a.B > 0 && a.B <= kBbMaxGrid, "attn_block_bwd: B=%d outside [1,%d] (feta_attn_block_bwd_blocks)", a.B, kBbMaxGrid);
  FETA_REQUIRE(a.N >= 1 && a.N <= 64 && a.M == a.B * a.N, "attn_block_bwd: N=%d outside [1,64] or M != B*N", a.N);
  FETA_REQUIRE(!a.y1 || (a.bn1 && a.g_sum && a.Gs > 0), "attn_block_bwd: y1 needs bn1, g_sum, Gs");
  FETA_REQUIRE(!a.sum_out || a.bn0, "attn_block_bwd: sum_out needs bn0");
  FETA_REQUIRE(aligned16(a.dy) && aligned16(a.qkv) && aligned16(a.out) && aligned16(a.x0) && aligned16(a.dx) &&
               aligned16(a.y1) && aligned16(a.dout2) && aligned16(a.g_sum) && aligned16(a.dx_b),
               "attn_block_bwd: tensors must be 16-byte aligned");
  const int nt = (a.N + 15) / 16;
  switch (nt) {
    case 1: return launch_block_bwd<1>(a, (hipStream_t)stream);
    case 2: return launch_block_bwd<2>(a, (hipStream_t)stream);
    case 3: return launch_block_bwd<3>(a, (hipStream_t)stream);
    default: return launch_block_bwd<4>(a, (hipStream_t)stream);
  }
}
