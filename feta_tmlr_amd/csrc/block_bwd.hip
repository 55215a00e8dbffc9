// Backward of the attention sub-block of one encoder layer, dX chain in ONE launch per layer
// (d = 64 = 4 heads x 16, N <= 48): BatchNorm-1 backward of the incoming gradient -> out_proj dX ->
// attention backward (dq, dk, dv) -> in_proj dX + residual gradient (+ the partial sums the previous
// layer's BatchNorm-2 backward needs).  Replaces the dX roles of two feta_rowlin_bwd_ex launches and
// feta_attn_bwd; the weight gradients of out_proj / in_proj are left to dW-only feta_rowlin_bwd_ex
// launches, which read this kernel's dqkv and the published (m1, m2).
// One workgroup per graph, one wave per head - the layout of attn_block_fwd_kernel (block.hip):
//   * W_out, W_in, the graph's gradient rows g0 = BNbwd(dy) [N,64], its q|k|v rows [N,192], pe_b and
//     the softmax statistics are staged once in LDS with 16-byte requests;
//   * dconcat = (degree * g0) W_out (+ the filter branch's gradient) is produced per 16 columns by
//     the wave that owns them and meets in an LDS tile, from which every head reads its slice in both
//     operand layouts;
//   * a head's wave runs the dq role over its query tiles and the dk/dv role over its key tiles with
//     the arithmetic of attn_bwd_dense_kernel (attn.hip) and writes dq|dk|dv over its own columns of
//     the q|k|v tile;
//   * wave w then owns columns 16w .. 16w+15 of dx = dqkv W_in + g0: the weight column slice sits in
//     registers, the dqkv rows are 16-byte LDS operands, the column sums need no cross-wave step.
#include "feta_abi_common.h"
#include "feta_rowops.h"

namespace feta {

typedef feta_attn_block_grad BwdArgs;  // include/feta_hip.h

constexpr int kBbD = 64, kBbH = 4, kBbDH = 16;
constexpr int kBbP = kBbD + 4;       // pitch of 64-float rows
constexpr int kBbPQ = 3 * kBbD + 4;  // pitch of the q|k|v rows

__host__ __device__ inline int block_bwd_lds_floats(int nt) {
  const int nr = 16 * nt;
  return kBbD * kBbP + 3 * kBbD * kBbP      // W_out, W_in
         + nr * kBbP + nr * kBbPQ + nr * kBbP  // g0 tile, q|k|v tile, dconcat tile
         + nr * (nr + 1)                      // pe
         + kBbH * nr * 2 + kBbH * nr          // softmax statistics, delta
         + 5 * kBbD + 2 * kBbD                // BatchNorm-1 backward parameters, mean / rstd of the input BN
         + reduce_scratch_floats(kBbD);
}

template <int NT>
__global__ __launch_bounds__(kRowThreads) void attn_block_bwd_kernel(BwdArgs a) {
  constexpr int D = kBbD, DH = kBbDH, H = kBbH, P = kBbP, PQ = kBbPQ, NR = 16 * NT, PEP = NR + 1;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, lq = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  const int n = a.n_real[b];
  float* Wo = feta_lds;           // [64][P]
  float* Wi = Wo + D * P;         // [192][P]
  float* Gs = Wi + 3 * D * P;     // [NR][P]   g0: gradient w.r.t. y1 (BatchNorm-1 backward applied)
  float* QK = Gs + NR * P;        // [NR][PQ]  q|k|v, later dq|dk|dv
  float* DO = QK + NR * PQ;       // [NR][P]   dconcat
  float* PE = DO + NR * P;        // [NR][PEP]
  float* ST = PE + NR * PEP;      // [H][NR][2] row max, row sum
  float* DL = ST + H * NR * 2;    // [H][NR]    delta = rowsum(dconcat * out)
  float* gv = DL + H * NR;        // [5][64] scale, mean, rstd, m1, m2 of BatchNorm 1
  float* ev = gv + 5 * D;         // [2][64] mean, rstd of the input BatchNorm (sums)
  float* scr = ev + 2 * D;        // finalize scratch
  const bool has_pe = a.pe != nullptr;
  const bool want_sums = a.sum_out != nullptr;
  auto grow = [&](int node) { return (int64_t)b * a.row_sb + (int64_t)min(node, a.N - 1) * a.row_sn; };

  // ---- requests ------------------------------------------------------------------------------------
  float4 dyv[NT], y1v[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int idx = tid + kRowThreads * i, node = idx >> 4, q = idx & 15;
    dyv[i] = *reinterpret_cast<const float4*>(a.dy + grow(node) * D + 4 * q);
    y1v[i] = *reinterpret_cast<const float4*>(a.y1 + grow(node) * D + 4 * q);
  }
  float4 qv[3 * NT];
#pragma unroll
  for (int i = 0; i < 3 * NT; ++i) {
    const int idx = tid + kRowThreads * i, node = idx / 48, q = idx - node * 48;
    qv[i] = *reinterpret_cast<const float4*>(a.qkv + grow(node) * 3 * D + 4 * q);
  }
  constexpr int PEI = (NR * NR + kRowThreads - 1) / kRowThreads;
  float pev[PEI];
#pragma unroll
  for (int i = 0; i < PEI; ++i) {
    const int idx = tid + kRowThreads * i, qq = idx / NR, kk = idx - qq * NR;
    const float v = has_pe ? a.pe[((int64_t)b * a.N + min(qq, a.N - 1)) * a.N + min(kk, a.N - 1)] : 1.0f;
    pev[i] = (idx < NR * NR && qq < a.N && kk < a.N) ? v : 0.0f;
  }
  constexpr int STI = (H * NR * 2 + kRowThreads - 1) / kRowThreads;
  float stv[STI];
#pragma unroll
  for (int i = 0; i < STI; ++i) {
    const int idx = tid + kRowThreads * i, hh = idx / (NR * 2), rem = idx - hh * NR * 2;
    const int qq = rem >> 1;
    const int hc = hh < H ? hh : H - 1;
    stv[i] = a.attn_stats[(((int64_t)b * H + hc) * a.N + min(qq, a.N - 1)) * 2 + (rem & 1)];
  }
  // per-lane operands of the later phases: degree of this lane's rows, out rows (delta), dout2 elements
  float rsv[NT];
  float4 ofv[NT], x0v[NT];
  float d2a[NT][4];   // dout2[node = 16t + 4g + r][16h + lq]
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t row = grow(16 * t + lq);
    rsv[t] = a.rowscale != nullptr ? a.rowscale[row] : 1.0f;
    ofv[t] = *reinterpret_cast<const float4*>(a.out + row * D + DH * h + 4 * g);
    // unconditional load (a conditionally-initialised float4 array is lowered to private memory):
    // without sums the values of dy are read and never used
    x0v[t] = *reinterpret_cast<const float4*>((want_sums ? a.x0 : a.dy) + row * D + DH * h + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      d2a[t][r] = a.dout2 != nullptr ? a.dout2[grow(16 * t + 4 * g + r) * D + DH * h + lq] : 0.0f;
  }
  {  // weights: W_out rows 0..63 then W_in rows 0..191 into the adjacent Wo | Wi regions, 16 per thread
    float4 wv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = tid + kRowThreads * i, r = idx >> 4, q = idx & 15;
      const float* src = r < D ? a.w_out + (int64_t)r * D : a.w_in + (int64_t)(r - D) * D;
      wv[i] = *reinterpret_cast<const float4*>(src + 4 * q);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = tid + kRowThreads * i;
      *reinterpret_cast<float4*>(Wo + (idx >> 4) * P + 4 * (idx & 15)) = wv[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 3 * NT; ++i) {
    const int idx = tid + kRowThreads * i, node = idx / 48, q = idx - node * 48;
    *reinterpret_cast<float4*>(QK + node * PQ + 4 * q) = qv[i];
  }
#pragma unroll
  for (int i = 0; i < PEI; ++i) {
    const int idx = tid + kRowThreads * i;
    if (idx < NR * NR) PE[(idx / NR) * PEP + idx % NR] = pev[i];
  }
#pragma unroll
  for (int i = 0; i < STI; ++i) {
    const int idx = tid + kRowThreads * i;
    if (idx < H * NR * 2) ST[idx] = stv[i];
  }
  // ---- BatchNorm-1 backward parameters: finalize (m1, m2), publish them and dgamma / dbeta -----------
  reduce_partials(a.g_sum, a.Gs, D, scr + 2 * D, scr);
  for (int c = tid; c < D; c += kRowThreads) {
    gv[c] = a.bn1[c];
    gv[D + c] = a.bn1[2 * D + c];
    gv[2 * D + c] = a.bn1[3 * D + c];
    gv[3 * D + c] = scr[c] / (float)a.M;
    gv[4 * D + c] = scr[D + c] / (float)a.M;
    ev[c] = want_sums ? a.bn0[2 * D + c] : 0.0f;
    ev[D + c] = want_sums ? a.bn0[3 * D + c] : 0.0f;
    if (b == 0) {
      a.dbeta[c] = scr[c];
      a.dgamma[c] = scr[D + c];
      a.fin_out[c] = gv[3 * D + c];
      a.fin_out[D + c] = gv[4 * D + c];
    }
  }
  __syncthreads();
  // ---- g0 = scale (dy - m1 - xhat m2), xhat = (y1 - mean) rstd --------------------------------------------
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int idx = tid + kRowThreads * i, node = idx >> 4, q = idx & 15;
    const float dd[4] = {dyv[i].x, dyv[i].y, dyv[i].z, dyv[i].w};
    const float yy[4] = {y1v[i].x, y1v[i].y, y1v[i].z, y1v[i].w};
    float v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 4 * q + s;
      const float xh = (yy[s] - gv[D + c]) * gv[2 * D + c];
      v[s] = node < a.N ? gv[c] * (dd[s] - gv[3 * D + c] - xh * gv[4 * D + c]) : 0.0f;
    }
    *reinterpret_cast<float4*>(Gs + node * P + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();

  // ---- dconcat[:, 16h .. 16h+15] = (degree * g0) W_out[:, 16h ..] (+ dout2): this wave's columns ----------
  {
    float woA[4][4];  // W_out[o = 16j + 4g + s][16h + lq]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) woA[j][s] = Wo[(16 * j + 4 * g + s) * P + DH * h + lq];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      Feat<D> gf;
      load_row<D>(gf, Gs + (16 * t + lq) * P, g, rsv[t]);
      f32x4 acc = zero4();
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma16(gf.f[j][s], woA[j][s], acc);  // (node 4g+r, column lq)
#pragma unroll
      for (int r = 0; r < 4; ++r) DO[(16 * t + 4 * g + r) * P + DH * h + lq] = acc[r] + d2a[t][r];
    }
  }
  __syncthreads();
  // ---- attention backward of head h (arithmetic of attn_bwd_dense_kernel) --------------------------------
  const int qo = DH * h, ko = D + DH * h, vo = 2 * D + DH * h;
  Feat<DH> qf[NT], kf[NT], vf[NT], dof[NT];
  float kb[NT][4], qb4[NT][4], dob[NT][4];
  float delta[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int rowl = 16 * t + lq;
    load_row<DH>(qf[t], QK + rowl * PQ + qo, g, a.scale);
    load_row<DH>(kf[t], QK + rowl * PQ + ko, g);
    load_row<DH>(vf[t], QK + rowl * PQ + vo, g);
    load_row<DH>(dof[t], DO + rowl * P + qo, g);
    if (rowl >= n) {
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[t].f[0][s] = vf[t].f[0][s] = 0.0f;
    }
    if (rowl >= a.N) {
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[t].f[0][s] = dof[t].f[0][s] = 0.0f;
    }
    float dl = dof[t].f[0][0] * ofv[t].x + dof[t].f[0][1] * ofv[t].y + dof[t].f[0][2] * ofv[t].z +
               dof[t].f[0][3] * ofv[t].w;
    dl += shfl_xor(dl, 16);
    dl += shfl_xor(dl, 32);
    delta[t] = dl;
    if (g == 0) DL[h * NR + rowl] = dl;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * t + 4 * g + r;
      kb[t][r] = rr < n ? QK[rr * PQ + ko + lq] : 0.0f;
      qb4[t][r] = rr < a.N ? QK[rr * PQ + qo + lq] * a.scale : 0.0f;
      dob[t][r] = rr < a.N ? DO[rr * P + qo + lq] : 0.0f;
    }
  }
  wave_lds_sync();  // DL of this head is read below by other lanes of the same wave
  // dq: S^T orientation (key 4g+r, query lq)
  f32x4 dq[NT];
#pragma unroll
  for (int qb = 0; qb < NT; ++qb) {
    dq[qb] = zero4();
    const int q = 16 * qb + lq;
    const float m = ST[(h * NR + q) * 2], z = ST[(h * NR + q) * 2 + 1];
    const float rinv = 1.0f / fmaxf(z, 1e-6f);
    const float dl = z < 1e-6f ? 0.0f : delta[qb];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (16 * kt >= n) continue;
      const f32x4 s = dot_rows<DH>(kf[kt], qf[qb], zero4());
      const f32x4 da = dot_rows<DH>(vf[kt], dof[qb], zero4());
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * g + r;
        const float p = key < n ? fast_exp(s[r] - m) * PE[q * PEP + key] * rinv : 0.0f;
        dq[qb] = mfma16(p * (da[r] - dl), kb[kt][r], dq[qb]);  // (query 4g+r, c lq)
      }
    }
  }
  // dk, dv: S orientation (query 4g+r, key lq)
  f32x4 dk[NT], dv[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) {
    dk[kt] = zero4();
    dv[kt] = zero4();
    if (16 * kt >= n) continue;
    const int key = 16 * kt + lq;
#pragma unroll
    for (int qb = 0; qb < NT; ++qb) {
      const f32x4 s = dot_rows<DH>(qf[qb], kf[kt], zero4());
      const f32x4 da = dot_rows<DH>(dof[qb], vf[kt], zero4());
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qb + 4 * g + r;
        const float m = ST[(h * NR + q) * 2], z = ST[(h * NR + q) * 2 + 1];
        const bool ok = q < a.N && key < n;
        const float p = ok ? fast_exp(s[r] - m) * PE[q * PEP + key] * (1.0f / fmaxf(z, 1e-6f)) : 0.0f;
        const float ds = p * (da[r] - (z < 1e-6f ? 0.0f : DL[h * NR + q]));
        dv[kt] = mfma16(p, dob[qb][r], dv[kt]);   // (key 4g+r, c lq)
        dk[kt] = mfma16(ds, qb4[qb][r], dk[kt]);
      }
    }
  }
  // dq | dk | dv over this head's columns of the q|k|v tile (its operands are all in registers)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * t + 4 * g + r;
      QK[rr * PQ + qo + lq] = dq[t][r] * a.scale;
      QK[rr * PQ + ko + lq] = dk[t][r];
      QK[rr * PQ + vo + lq] = dv[t][r];
    }
  __syncthreads();

  // ---- dqkv to HBM (whole rows) and dx = dqkv W_in + g0: wave w owns columns 16w .. 16w+15 ---------------
#pragma unroll
  for (int i = 0; i < 3 * NT; ++i) {
    const int idx = tid + kRowThreads * i, node = idx / 48, q = idx - node * 48;
    if (node < a.N)
      *reinterpret_cast<float4*>(a.dqkv + grow(node) * 3 * D + 4 * q) =
          *reinterpret_cast<const float4*>(QK + node * PQ + 4 * q);
  }
  {
    float wA[12][4];  // W_in[j = 16jj + 4g + s][16h + lq]
#pragma unroll
    for (int jj = 0; jj < 12; ++jj)
#pragma unroll
      for (int s = 0; s < 4; ++s) wA[jj][s] = Wi[(16 * jj + 4 * g + s) * P + DH * h + lq];
    float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int node = 16 * t + lq;
      const bool rok = node < a.N;
      Feat<3 * D> gf;
      load_row<3 * D>(gf, QK + node * PQ, g);
      f32x4 acc = zero4();
#pragma unroll
      for (int jj = 0; jj < 12; ++jj)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma16(wA[jj][s], gf.f[jj][s], acc);  // (k = 16h + 4g + r, node lq)
      const float4 g0 = *reinterpret_cast<const float4*>(Gs + node * P + DH * h + 4 * g);
      const float v[4] = {acc[0] + g0.x, acc[1] + g0.y, acc[2] + g0.z, acc[3] + g0.w};
      if (rok) *reinterpret_cast<float4*>(a.dx + grow(node) * D + DH * h + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
      if (want_sums) {
        const float xx[4] = {x0v[t].x, x0v[t].y, x0v[t].z, x0v[t].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = DH * h + 4 * g + r;
          const float xh = (xx[r] - ev[c]) * ev[D + c];
          const float t1 = rok ? v[r] : 0.0f;
          s1[r] += t1;
          s2[r] += t1 * xh;
        }
      }
    }
    if (want_sums) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = row16_sum(s1[r]), t2 = row16_sum(s2[r]);
        if (lq == 0) {
          a.sum_out[(int64_t)b * 2 * D + DH * h + 4 * g + r] = t1;
          a.sum_out[(int64_t)b * 2 * D + D + DH * h + 4 * g + r] = t2;
        }
      }
    }
  }
}

template <int NT>
int launch_block_bwd(const BwdArgs& a, hipStream_t stream) {
  const size_t lds = sizeof(float) * block_bwd_lds_floats(NT);
  auto kern = attn_block_bwd_kernel<NT>;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(kRowThreads), lds, stream, a);
  return check_launch("feta_attn_block_bwd");
}

}  // namespace feta

using namespace feta;

extern "C" int feta_attn_block_bwd_supported(int N, int d_model, int heads) {
  return (d_model == kBbD && heads == kBbH && N >= 1 && N <= 48) ? 1 : 0;
}

extern "C" int feta_attn_block_bwd(const feta_attn_block_grad* d, feta_stream_t stream) {
  FETA_REQUIRE(d != nullptr, "attn_block_bwd: null descriptor");
  const BwdArgs& a = *d;
  FETA_REQUIRE(a.dy && a.y1 && a.bn1 && a.g_sum && a.fin_out && a.dgamma && a.dbeta && a.w_out && a.w_in &&
                   a.qkv && a.out && a.n_real && a.attn_stats && a.dqkv && a.dx,
               "attn_block_bwd: null pointer");
  FETA_REQUIRE(a.B > 0 && a.N >= 1 && a.N <= 48, "attn_block_bwd: N=%d outside [1,48]", a.N);
  FETA_REQUIRE(a.M == a.B * a.N && a.Gs > 0, "attn_block_bwd: M=%d is not B*N, or Gs <= 0", a.M);
  FETA_REQUIRE(!a.sum_out || (a.x0 && a.bn0), "attn_block_bwd: sum_out needs x0 and bn0");
  FETA_REQUIRE(aligned16(a.dy) && aligned16(a.y1) && aligned16(a.g_sum) && aligned16(a.w_out) && aligned16(a.w_in) &&
                   aligned16(a.qkv) && aligned16(a.out) && aligned16(a.dqkv) && aligned16(a.dx) && aligned16(a.x0),
               "attn_block_bwd: tensors must be 16-byte aligned");
  const int nt = (a.N + 15) / 16;
  switch (nt) {
    case 1: return launch_block_bwd<1>(a, (hipStream_t)stream);
    case 2: return launch_block_bwd<2>(a, (hipStream_t)stream);
    default: return launch_block_bwd<3>(a, (hipStream_t)stream);
  }
}
