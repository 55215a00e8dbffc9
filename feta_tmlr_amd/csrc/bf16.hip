// bf16 storage path of the attention core (A1) and the eigenbasis filter (A3): BASELINE configs 3 and 5
// ("8 x MI355X ... bf16", "ogbg-molhiv ... bf16, MFMA QK^T path").  The reference has no reduced-precision
// mode (no AMP in experiments/run_transformer_gengcn.py:115-164): parity of this path is "within a stated
// bf16 tolerance of the fp64 oracle", the fp32 kernels stay the reference arithmetic.
//
// Same work decomposition and operand orientations as the general kernels of attn.hip / filter.hip (one wave
// per (graph, head, 16-row block) resp. per (graph, head)); written against a storage type (feta_bf16.h), so
// q/k/v, pe, out, attn, x, U, the per-block weights and every gradient move as 2-byte elements, the QK^T,
// P.V, dS.K, U^T.X, U.Y contractions run as v_mfma_f32_16x16x16_bf16, and the softmax statistics, delta,
// lambda, t_k(lambda) and all accumulators are fp32.
#include <cmath>

#include "feta_abi_common.h"
#include "feta_bf16.h"

namespace feta {

constexpr int kLWaves = 4;

template <class T>
struct AttnArgsT {
  const T* q;
  const T* k;
  const T* v;
  const T* pe;
  const int32_t* n_real;
  const T* out;
  const T* dout;
  const float* stats_in;
  T* out_w;
  T* attn;
  float* stats;
  float* delta;
  T* dq;
  T* dk;
  T* dv;
  int64_t qsb, qsn, osb, osn;
  float scale;
  int B, N, H, NB, total;
  // attention-probability dropout (A1 step (6)): keep-threshold on 32 random bits per probability,
  // Philox4x32-10 keyed by `seed`, counter = (index of the 4-key group of the probability, `offset`): the mask is
  // a pure function of (seed, offset, b, h, query, key), regenerated in backward - no mask tensor
  unsigned drop_thresh;   // keep iff bits >= drop_thresh; 0 = no dropout
  float drop_scale;       // 1 / (1 - p)
  unsigned seed_lo, seed_hi, off_lo, off_hi;
  // device-resident (seed, offset) (feta_attn_*_drop_dev): when given, the mask is keyed by dstate[0] and counted from
  // dstate[1] + (off_hi, off_lo) - read at RUN time, so a captured hipGraph draws a fresh mask on every replay
  const unsigned long long* dstate;
  // stabilisation of the exponent (stab = {rowmax, clamp5}): 0 = exp(s - rowmax) (upstream GraphiT), 1 = the
  // in-tree witnesses' form exp(clamp(s, -5, 5)) (LSPE/layers/graphit_gt_layer.py:39-43,
  // LPE/layers/graph_transformer_spectra_layer.py:239-243) - no row maximum (the stored one is 0), no gradient through
  // a clamped score
  int clamp5;
};

// (kernel arguments are a by-value copy: the device-resident key is patched into it once per wave, scalar loads)
template <class A>
__device__ __forceinline__ void resolve_drop_state(A& a) {
  if (a.dstate != nullptr) {
    const unsigned long long seed = a.dstate[0];
    const unsigned long long off = a.dstate[1] + (((unsigned long long)a.off_hi << 32) | a.off_lo);
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
    a.off_lo = (unsigned)off; a.off_hi = (unsigned)(off >> 32);
  }
}

__device__ __forceinline__ unsigned mulhi32(unsigned a, unsigned b) {
  return (unsigned)(((unsigned long long)a * (unsigned long long)b) >> 32);
}
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): 4 x 32 random bits per counter
__device__ __forceinline__ void philox4(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                        unsigned (&o)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// keep-scales (0 or 1/(1-p)) of the probabilities (query q, keys 4 kg .. 4 kg + 3) of block bh
template <class A>
__device__ __forceinline__ void drop_scales(const A& a, int bh, int q, int kg, float (&ms)[4]) {
  const unsigned long long idx = ((unsigned long long)bh * a.N + q) * (unsigned long long)((a.N + 3) >> 2) + kg;
  unsigned bits[4];
  philox4((unsigned)idx, (unsigned)(idx >> 32), a.off_lo, a.off_hi, a.seed_lo, a.seed_hi, bits);
#pragma unroll
  for (int r = 0; r < 4; ++r) ms[r] = bits[r] >= a.drop_thresh ? a.drop_scale : 0.0f;
}

// ---- forward: one wave per (b, h, 16-query block), S^T in registers ------------------------------------------
template <class T, int DH, int KT_MAX>
__global__ __launch_bounds__(64 * kLWaves) void attn_fwd_lp_kernel(AttnArgsT<T> a) {
  resolve_drop_state(a);
  constexpr int CT = Feat<DH>::CT;
  constexpr int KP = 16 * KT_MAX + 1;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kLWaves + wave_id();
  if (item >= a.total) return;
  const int qb = item % a.NB;
  const int bh = item / a.NB;
  const int h = bh % a.H, b = bh / a.H;
  const int n = a.n_real[b];
  const int q0 = 16 * qb;
  const int KT = (n + 15) >> 4;
  const int q = q0 + lq;
  const int nm1 = max(n - 1, 0), qc = min(q, a.N - 1);
  Feat<DH> qf;
  load_row_t<T, DH>(qf, tok_row_t(a.q, a.qsb, a.qsn, b, qc, h, DH), q < a.N, g, a.scale);

  f32x4 acc[KT_MAX];   // S^T: acc[kt][r] <-> key 16kt + 4g + r, query q
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    if (kt < KT) {
      const int key = 16 * kt + lq;
      Feat<DH> kf;
      load_row_t<T, DH>(kf, tok_row_t(a.k, a.qsb, a.qsn, b, min(key, nm1), h, DH), key < n, g);
      acc[kt] = dot_rows_t<T, DH>(kf, qf, zero4());
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    if (kt < KT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (a.clamp5) acc[kt][r] = fminf(fmaxf(acc[kt][r], -5.0f), 5.0f);
        if (16 * kt + 4 * g + r < n) m = fmaxf(m, acc[kt][r]);
      }
    }
  }
  m = fmaxf(m, shfl_xor(m, 16));
  m = fmaxf(m, shfl_xor(m, 32));
  if (a.clamp5) m = 0.0f;   // (wave-uniform flag; the shuffles above stay unconditional)

  const bool has_pe = a.pe != nullptr;
  const T* pe_row = has_pe ? a.pe + ((int64_t)b * a.N + qc) * a.N : nullptr;
  float z = 0.0f;
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    if (kt < KT) {
      float pv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pv[r] = has_pe ? Num<T>::ld(pe_row + min(16 * kt + 4 * g + r, nm1)) : 1.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * g + r;
        const float e = key < n ? fast_exp(acc[kt][r] - m) * pv[r] : 0.0f;
        acc[kt][r] = e;
        z += e;
      }
    }
  }
  z += shfl_xor(z, 16);
  z += shfl_xor(z, 32);
  const float rinv = 1.0f / fmaxf(z, 1e-6f);
  if (g == 0 && q < a.N) {
    float* st = a.stats + ((int64_t)bh * a.N + q) * 2;
    st[0] = m;
    st[1] = z;
  }

  // out = P . V: contraction over keys = accumulator rows; V[key 4g+r][c = lq] as the B operand
  f32x4 o[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) o[ct] = zero4();
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    if (kt < KT) {
      float pa[4];
      float vb[CT][4];
      float ms[4] = {1.0f, 1.0f, 1.0f, 1.0f};
      if (a.drop_thresh != 0u) drop_scales(a, bh, qc, 4 * kt + g, ms);   // (written to attn dropped, as nn.MultiheadAttention)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[kt][r] *= rinv * ms[r];
        pa[r] = acc[kt][r];
        const int key = 16 * kt + 4 * g + r;
        const T* vrow = tok_row_t(a.v, a.qsb, a.qsn, b, min(key, nm1), h, DH);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int c = 16 * ct + lq;
          const float vv = Num<T>::ld(vrow + (c < DH ? c : 0));
          vb[ct][r] = (key < n && c < DH) ? vv : 0.0f;
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) o[ct] = Num<T>::mma4(pa, vb[ct], o[ct]);
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * ct + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 4 * g + r;
      if (qq < a.N && c < DH) Num<T>::st(tok_row_t(a.out_w, a.osb, a.osn, b, qq, h, DH) + c, o[ct][r]);
    }
  }

  // attn[b,h,q0:q0+16,:] is one contiguous run: stage the block in LDS (fp32), store coalesced
  if (a.attn != nullptr) {
    float* st = feta_lds + wave_id() * 16 * KP;
#pragma unroll
    for (int kt = 0; kt < KT_MAX; ++kt) {
      if (kt < KT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) st[lq * KP + 16 * kt + 4 * g + r] = acc[kt][r];
      }
    }
    wave_lds_sync();
    const int rows = min(16, a.N - q0);
    const int cols = 16 * KT;
    T* dst = a.attn + ((int64_t)bh * a.N + q0) * a.N;
    for (int idx = lane; idx < rows * a.N; idx += 64) {
      const int qq = idx / a.N, kk = idx - qq * a.N;
      Num<T>::st(dst + idx, kk < cols ? st[qq * KP + kk] : 0.0f);
    }
  }
}

// ---- dq for one 16-query block (+ delta = rowsum(dout * out)) -------------------------------------------------
template <class T, int DH>
__global__ __launch_bounds__(64 * kLWaves) void attn_bwd_dq_lp_kernel(AttnArgsT<T> a) {
  resolve_drop_state(a);
  constexpr int CT = Feat<DH>::CT;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kLWaves + wave_id();
  if (item >= a.total) return;
  const int qb = item % a.NB;
  const int bh = item / a.NB;
  const int h = bh % a.H, b = bh / a.H;
  const int n = a.n_real[b];
  const int q0 = 16 * qb;
  const int KT = (n + 15) >> 4;
  const int q = q0 + lq, qc = min(q, a.N - 1);
  const bool qok = q < a.N;

  Feat<DH> qf, dof, of;
  load_row_t<T, DH>(qf, tok_row_t(a.q, a.qsb, a.qsn, b, qc, h, DH), qok, g, a.scale);
  load_row_t<T, DH>(dof, tok_row_t(a.dout, a.osb, a.osn, b, qc, h, DH), qok, g);
  load_row_t<T, DH>(of, tok_row_t(a.out, a.osb, a.osn, b, qc, h, DH), qok, g);
  float delta = 0.0f;
#pragma unroll
  for (int j = 0; j < Feat<DH>::NJ; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) delta += dof.f[j][s] * of.f[j][s];
  delta += shfl_xor(delta, 16);
  delta += shfl_xor(delta, 32);
  if (g == 0 && qok) a.delta[(int64_t)bh * a.N + q] = delta;

  float m = 0.0f, z = 1.0f;
  {
    const float* st = a.stats_in + ((int64_t)bh * a.N + qc) * 2;
    m = st[0];
    z = st[1];
  }
  const float rinv = 1.0f / fmaxf(z, 1e-6f);
  if (z < 1e-6f) delta = 0.0f;  // clamp active: the normaliser is a constant
  f32x4 dq[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) dq[ct] = zero4();
  const int nm1 = max(n - 1, 0);
  const bool has_pe = a.pe != nullptr;
  const T* pe_c = has_pe ? a.pe + ((int64_t)b * a.N + qc) * a.N : nullptr;
  for (int kt = 0; kt < KT; ++kt) {
    const int krow = 16 * kt + lq;
    Feat<DH> kf, vf;
    load_row_t<T, DH>(kf, tok_row_t(a.k, a.qsb, a.qsn, b, min(krow, nm1), h, DH), krow < n, g);
    load_row_t<T, DH>(vf, tok_row_t(a.v, a.qsb, a.qsn, b, min(krow, nm1), h, DH), krow < n, g);
    float pv[4], kb[CT][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * g + r;
      const int kc = min(key, nm1);
      pv[r] = has_pe ? Num<T>::ld(pe_c + kc) : 1.0f;
      const T* kr = tok_row_t(a.k, a.qsb, a.qsn, b, kc, h, DH);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = 16 * ct + lq;
        const float kv = Num<T>::ld(kr + (c < DH ? c : 0));
        kb[ct][r] = (key < n && c < DH) ? kv : 0.0f;
      }
    }
    const f32x4 s = dot_rows_t<T, DH>(kf, qf, zero4());    // scores^T
    const f32x4 da = dot_rows_t<T, DH>(vf, dof, zero4());  // (dout . v^T)^T
    float ds[4];
    float ms[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (a.drop_thresh != 0u) drop_scales(a, bh, qc, 4 * kt + g, ms);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * g + r;
      const float sc = a.clamp5 ? fminf(fmaxf(s[r], -5.0f), 5.0f) : s[r];
      const float p = (key < n && qok) ? fast_exp(sc - m) * pv[r] * rinv : 0.0f;
      ds[r] = p * (da[r] * ms[r] - delta);   // d(dropped probability) -> d(probability): the same keep-scale
      if (a.clamp5 && sc != s[r]) ds[r] = 0.0f;   // a clamped score passes no gradient
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dq[ct] = Num<T>::mma4(ds, kb[ct], dq[ct]);
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * ct + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 4 * g + r;
      if (qq < a.N && c < DH) Num<T>::st(tok_row_t(a.dq, a.qsb, a.qsn, b, qq, h, DH) + c, dq[ct][r] * a.scale);
    }
  }
}

// ---- dk, dv for one 16-key block (rows = queries, column = key) ---------------------------------------------------
template <class T, int DH>
__global__ __launch_bounds__(64 * kLWaves) void attn_bwd_dkdv_lp_kernel(AttnArgsT<T> a) {
  resolve_drop_state(a);
  constexpr int CT = Feat<DH>::CT;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kLWaves + wave_id();
  if (item >= a.total) return;
  const int kb = item % a.NB;
  const int bh = item / a.NB;
  const int h = bh % a.H, b = bh / a.H;
  const int n = a.n_real[b];
  const int key = 16 * kb + lq;
  const bool kok = key < n;

  f32x4 dk[CT], dv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    dk[ct] = zero4();
    dv[ct] = zero4();
  }
  if (16 * kb < n) {
    const int keyc = min(key, max(n - 1, 0));
    Feat<DH> kf, vf;
    load_row_t<T, DH>(kf, tok_row_t(a.k, a.qsb, a.qsn, b, keyc, h, DH), kok, g);
    load_row_t<T, DH>(vf, tok_row_t(a.v, a.qsb, a.qsn, b, keyc, h, DH), kok, g);
    const bool has_pe = a.pe != nullptr;
    for (int qb = 0; qb < a.NB; ++qb) {
      const int qrow = 16 * qb + lq, qrc = min(qrow, a.N - 1);
      Feat<DH> qf, dof;
      load_row_t<T, DH>(qf, tok_row_t(a.q, a.qsb, a.qsn, b, qrc, h, DH), qrow < a.N, g, a.scale);
      load_row_t<T, DH>(dof, tok_row_t(a.dout, a.osb, a.osn, b, qrc, h, DH), qrow < a.N, g);
      float sm[4], sz[4], sd[4], pv[4], dob[CT][4], qbv[CT][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = 16 * qb + 4 * g + r, qqc = min(qq, a.N - 1);
        const float* st = a.stats_in + ((int64_t)bh * a.N + qqc) * 2;
        sm[r] = st[0];
        sz[r] = st[1];
        sd[r] = a.delta[(int64_t)bh * a.N + qqc];
        pv[r] = has_pe ? Num<T>::ld(a.pe + ((int64_t)b * a.N + qqc) * a.N + keyc) : 1.0f;
        const T* dorow = tok_row_t(a.dout, a.osb, a.osn, b, qqc, h, DH);
        const T* qrow_p = tok_row_t(a.q, a.qsb, a.qsn, b, qqc, h, DH);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int c = 16 * ct + lq, cc = c < DH ? c : 0;
          const bool ok = qq < a.N && c < DH;
          const float dv_ = Num<T>::ld(dorow + cc), qv_ = Num<T>::ld(qrow_p + cc);
          dob[ct][r] = ok ? dv_ : 0.0f;
          qbv[ct][r] = ok ? qv_ * a.scale : 0.0f;
        }
      }
      const f32x4 s = dot_rows_t<T, DH>(qf, kf, zero4());    // scores: row = query 4g+r, col = key
      const f32x4 da = dot_rows_t<T, DH>(dof, vf, zero4());  // dout . v^T
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = 16 * qb + 4 * g + r;
        const float zz = sz[r];
        float mk = 1.0f;
        if (a.drop_thresh != 0u) {   // this lane's key of query qq: element key & 3 of its 4-key group
          float ms[4];
          drop_scales(a, bh, min(qq, a.N - 1), keyc >> 2, ms);
          mk = (keyc & 3) == 0 ? ms[0] : ((keyc & 3) == 1 ? ms[1] : ((keyc & 3) == 2 ? ms[2] : ms[3]));
        }
        const float sc = a.clamp5 ? fminf(fmaxf(s[r], -5.0f), 5.0f) : s[r];
        const float pr = (qq < a.N && kok) ? fast_exp(sc - sm[r]) * pv[r] * (1.0f / fmaxf(zz, 1e-6f)) : 0.0f;
        ds[r] = pr * (da[r] * mk - (zz < 1e-6f ? 0.0f : sd[r]));
        if (a.clamp5 && sc != s[r]) ds[r] = 0.0f;
        p[r] = pr * mk;              // dV takes the dropped probabilities
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        dv[ct] = Num<T>::mma4(p, dob[ct], dv[ct]);
        dk[ct] = Num<T>::mma4(ds, qbv[ct], dk[ct]);
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * ct + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = 16 * kb + 4 * g + r;
      if (kk < a.N && c < DH) {
        Num<T>::st(tok_row_t(a.dk, a.qsb, a.qsn, b, kk, h, DH) + c, dk[ct][r]);
        Num<T>::st(tok_row_t(a.dv, a.qsb, a.qsn, b, kk, h, DH) + c, dv[ct][r]);
      }
    }
  }
}

// ---- eigenbasis filter ---------------------------------------------------------------------------------------
template <class T>
struct FilterArgsT {
  const T* x;
  const T* u;
  const float* lam;
  const T* coeff;
  const float* bias;
  const int32_t* n_real;
  const T* dy;
  T* y;
  T* dx;
  T* dcoeff;
  float* dbias_part;
  int64_t xsb, xsn, ysb, ysn;
  int B, N, H, P, K;
  int share;
  int total;
};

constexpr int kLMaxOrder = 8;
__device__ __forceinline__ void cheb_poly_lp(float lam, int P, float (&t)[kLMaxOrder]) {
  t[0] = 1.0f;
  t[1] = lam;
#pragma unroll
  for (int k = 2; k < kLMaxOrder; ++k) t[k] = (k < P) ? 2.0f * lam * t[k - 1] - t[k - 2] : 0.0f;
}
__device__ __forceinline__ float cheb_zero_lp(int k) { return (k & 1) ? 0.0f : ((k & 2) ? -1.0f : 1.0f); }

// acc-layout tile of a token tensor: rows row0 + 4g + r (< nvalid), column 16ct + lq
template <class T, int DH>
__device__ __forceinline__ f32x4 load_acc_t(const T* p, int64_t sb, int64_t sn, int b, int h, int row0, int nvalid,
                                            int ct, int lq, int g) {
  f32x4 v;
  const int c = 16 * ct + lq;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int node = row0 + 4 * g + r;
    const float x = Num<T>::ld(tok_row_t(p, sb, sn, b, min(node, max(nvalid - 1, 0)), h, DH) + (c < DH ? c : 0));
    v[r] = (node < nvalid && c < DH) ? x : 0.0f;
  }
  return v;
}
template <class T, int DH>
__device__ __forceinline__ void store_acc_t(T* p, int64_t sb, int64_t sn, int b, int h, int row0, int nrows, int ct,
                                            int lq, int g, const f32x4& v) {
  const int c = 16 * ct + lq;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int node = row0 + 4 * g + r;
    if (node < nrows && c < DH) Num<T>::st(tok_row_t(p, sb, sn, b, node, h, DH) + c, v[r]);
  }
}
// W_k[c = 16j+4g+s][c' = 16ct+lq], s = 0..3 (B operand of X . W_k)
template <class T, int DH>
__device__ __forceinline__ void w_b4(const T* w, int k, int j, int ct, int lq, int g, float (&o)[4]) {
  const int cp = 16 * ct + lq;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = 16 * j + 4 * g + s;
    const float v = Num<T>::ld(w + (k * DH + (c < DH ? c : 0)) * DH + (cp < DH ? cp : 0));
    o[s] = (c < DH && cp < DH) ? v : 0.0f;
  }
}
// W_k[c = 16ct+lq][c' = 16j+4g .. +3] (operand of dY . W_k^T)
template <class T, int DH>
__device__ __forceinline__ void w_row4_t(const T* w, int k, int ct, int j, int lq, int g, float (&o)[4]) {
  const int c = 16 * ct + lq, cp = 16 * j + 4 * g;
  Num<T>::ld4(w + (k * DH + (c < DH ? c : 0)) * DH + (cp < DH ? cp : 0), o);
  const bool ok = c < DH && cp < DH;
#pragma unroll
  for (int s = 0; s < 4; ++s) o[s] = ok ? o[s] : 0.0f;
}
// U[node][e] element
template <class T>
__device__ __forceinline__ float u_at(const T* U, int K, int node, int e, int n) {
  const float v = Num<T>::ld(U + (int64_t)min(node, max(n - 1, 0)) * K + min(e, K - 1));
  return (node < n && e < K) ? v : 0.0f;
}

template <class T, int DH, int ET_MAX>
__global__ __launch_bounds__(64 * kLWaves) void spec_fwd_lp_kernel(FilterArgsT<T> a) {
  constexpr int CT = Feat<DH>::CT, NJ = Feat<DH>::NJ;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kLWaves + wave_id();
  if (item >= a.total) return;
  const int h = item % a.H, b = item / a.H;
  const int n = a.n_real[b];
  const T* w = a.coeff + ((int64_t)h * a.B + b) * a.P * DH * DH;
  const int NT = (n + 15) >> 4;
  const int NTall = (a.N + 15) >> 4;
  const bool graph = a.share || h == 0;   // heads >= 1 of the reference-literal mode: Lhat = 0
  const int ET = (a.K + 15) >> 4;
  const T* U = a.u + (int64_t)b * a.N * a.K;
  const float* lam = a.lam + (int64_t)b * a.K;

  f32x4 yt[ET_MAX][CT];   // Ytil[e][c']
  if (graph) {
    // (1) Xtil^T[c][e] = sum_node X[node][c] U[node][e]
    f32x4 xtT[CT][ET_MAX];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) xtT[ct][et] = zero4();
    for (int nt = 0; nt < NT; ++nt) {
      float xa[CT][4], uu[ET_MAX][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nd = 16 * nt + 4 * g + r;
        const T* row = tok_row_t(a.x, a.xsb, a.xsn, b, min(nd, n - 1), h, DH);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int c = 16 * ct + lq;
          const float v = Num<T>::ld(row + (c < DH ? c : 0));
          xa[ct][r] = (nd < n && c < DH) ? v : 0.0f;
        }
#pragma unroll
        for (int et = 0; et < ET_MAX; ++et) uu[et][r] = u_at(U, a.K, nd, 16 * et + lq, n);
      }
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et)
        if (et < ET) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) xtT[ct][et] = Num<T>::mma4(xa[ct], uu[et], xtT[ct][et]);
        }
    }
    // (2) Ytil[e][c'] = sum_k t_k(lam_e) sum_c Xtil[e][c] W_k[c][c']
#pragma unroll
    for (int et = 0; et < ET_MAX; ++et) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) yt[et][ct] = zero4();
      if (et < ET) {
        const int e = 16 * et + lq;
        float tk[kLMaxOrder];
        cheb_poly_lp(e < a.K ? lam[e] : 0.0f, a.P, tk);
#pragma unroll
        for (int k = 0; k < kLMaxOrder; ++k) {
          if (k < a.P) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              float av[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) av[r] = xtT[ct][et][r] * tk[k];
#pragma unroll
              for (int c2 = 0; c2 < CT; ++c2) {
                float wv[4];
                w_b4<T, DH>(w, k, ct, c2, lq, g, wv);
                yt[et][c2] = Num<T>::mma4(av, wv, yt[et][c2]);
              }
            }
          }
        }
      }
    }
  }
  // (3) Y = U Ytil + bias   (no graph: Y = X sum_k T_k(0) W_k + bias)
  for (int nt = 0; nt < NTall; ++nt) {
    f32x4 y[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) y[ct] = zero4();
    if (nt < NT) {
      const int node = 16 * nt + lq;
      if (graph) {
#pragma unroll
        for (int et = 0; et < ET_MAX; ++et) {
          if (et < ET) {
            float ua[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ua[r] = u_at(U, a.K, node, 16 * et + 4 * g + r, n);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const float yv[4] = {yt[et][ct][0], yt[et][ct][1], yt[et][ct][2], yt[et][ct][3]};
              y[ct] = Num<T>::mma4(ua, yv, y[ct]);
            }
          }
        }
      } else {
        Feat<DH> xf;
        load_row_t<T, DH>(xf, tok_row_t(a.x, a.xsb, a.xsn, b, min(node, n - 1), h, DH), node < n, g);
        for (int k = 0; k < a.P; k += 2) {
          const float sgn = cheb_zero_lp(k);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            f32x4 zt = zero4();
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              float wv[4];
              w_b4<T, DH>(w, k, j, ct, lq, g, wv);
              zt = Num<T>::mma4(xf.f[j], wv, zt);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) y[ct][r] += sgn * zt[r];
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
      store_acc_t<T, DH>(a.y, a.ysb, a.ysn, b, h, 16 * nt, a.N, ct, lq, g, y[ct]);
    }
  }
}

template <class T, int DH, int ET_MAX>
__global__ __launch_bounds__(64 * kLWaves) void spec_bwd_lp_kernel(FilterArgsT<T> a) {
  constexpr int CT = Feat<DH>::CT, NJ = Feat<DH>::NJ;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kLWaves + wave_id();
  if (item >= a.total) return;
  const int h = item % a.H, b = item / a.H;
  const int n = a.n_real[b];
  const int64_t blk = (int64_t)h * a.B + b;
  const T* w = a.coeff + blk * a.P * DH * DH;
  T* dw = a.dcoeff + blk * a.P * DH * DH;
  const int NT = (n + 15) >> 4;
  const int NTall = (a.N + 15) >> 4;
  const bool graph = a.share || h == 0;
  const int ET = (a.K + 15) >> 4;
  const T* U = a.u + (int64_t)b * a.N * a.K;
  const float* lam = a.lam + (int64_t)b * a.K;
  f32x4 dbias[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) dbias[ct] = zero4();

  if (!graph) {
    // dW_k = T_k(0) X^T dY; dX = dY (sum_k T_k(0) W_k)^T
    for (int k = 0; k < a.P; ++k) {
      const float sgn = cheb_zero_lp(k);
      f32x4 acc[CT][CT];
#pragma unroll
      for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
        for (int c2 = 0; c2 < CT; ++c2) acc[c1][c2] = zero4();
      if (sgn != 0.0f || k == 0) {
        for (int nt = 0; nt < NT; ++nt) {
          f32x4 xa[CT], dyb[CT];
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            xa[ct] = load_acc_t<T, DH>(a.x, a.xsb, a.xsn, b, h, 16 * nt, n, ct, lq, g);
            dyb[ct] = load_acc_t<T, DH>(a.dy, a.ysb, a.ysn, b, h, 16 * nt, n, ct, lq, g);
            if (k == 0) {
#pragma unroll
              for (int r = 0; r < 4; ++r) dbias[ct][r] += dyb[ct][r];
            }
          }
#pragma unroll
          for (int c1 = 0; c1 < CT; ++c1) {
            f32x4 xs;
#pragma unroll
            for (int r = 0; r < 4; ++r) xs[r] = sgn * xa[c1][r];
#pragma unroll
            for (int c2 = 0; c2 < CT; ++c2) acc[c1][c2] = mma_acc<T>(xs, dyb[c2], acc[c1][c2]);
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
            if (c < DH && cp < DH) Num<T>::st(dw + (k * DH + c) * DH + cp, acc[c1][c2][r]);
          }
    }
    for (int nt = 0; nt < NTall; ++nt) {
      f32x4 dx[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) dx[ct] = zero4();
      if (nt < NT) {
        const int node = 16 * nt + lq;
        Feat<DH> dyf;
        load_row_t<T, DH>(dyf, tok_row_t(a.dy, a.ysb, a.ysn, b, min(node, n - 1), h, DH), node < n, g);
        for (int k = 0; k < a.P; k += 2) {
          const float sgn = cheb_zero_lp(k);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            f32x4 gk = zero4();
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              float wv[4];
              w_row4_t<T, DH>(w, k, ct, j, lq, g, wv);
              gk = Num<T>::mma4(dyf.f[j], wv, gk);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dx[ct][r] += sgn * gk[r];   // acc layout [node 4g+r][c = 16ct+lq]
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) store_acc_t<T, DH>(a.dx, a.xsb, a.xsn, b, h, 16 * nt, a.N, ct, lq, g, dx[ct]);
    }
  } else {
    // (1) Xtil = U^T X, dYtil = U^T dY (acc layout [e][c]), dYtil^T ([c'][e]); dbias partial
    f32x4 xt[ET_MAX][CT], dyt[ET_MAX][CT], dytT[CT][ET_MAX];
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
        xb[ct] = load_acc_t<T, DH>(a.x, a.xsb, a.xsn, b, h, 16 * nt, n, ct, lq, g);
        dyb[ct] = load_acc_t<T, DH>(a.dy, a.ysb, a.ysn, b, h, 16 * nt, n, ct, lq, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) dbias[ct][r] += dyb[ct][r];
      }
#pragma unroll
      for (int et = 0; et < ET_MAX; ++et) {
        if (et < ET) {
          f32x4 ua;
#pragma unroll
          for (int r = 0; r < 4; ++r) ua[r] = u_at(U, a.K, 16 * nt + 4 * g + r, 16 * et + lq, n);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            xt[et][ct] = mma_acc<T>(ua, xb[ct], xt[et][ct]);
            dyt[et][ct] = mma_acc<T>(ua, dyb[ct], dyt[et][ct]);
            dytT[ct][et] = mma_acc<T>(dyb[ct], ua, dytT[ct][et]);
          }
        }
      }
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
          f32x4 tke;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = 16 * et + 4 * g + r;
            float tk[kLMaxOrder];
            cheb_poly_lp(e < a.K ? lam[e] : 0.0f, a.P, tk);
            float t = 0.0f;
#pragma unroll
            for (int kk = 0; kk < kLMaxOrder; ++kk)
              if (kk == k) t = tk[kk];
            tke[r] = t;
          }
#pragma unroll
          for (int c1 = 0; c1 < CT; ++c1) {
            f32x4 xs;
#pragma unroll
            for (int r = 0; r < 4; ++r) xs[r] = xt[et][c1][r] * tke[r];
#pragma unroll
            for (int c2 = 0; c2 < CT; ++c2) acc[c1][c2] = mma_acc<T>(xs, dyt[et][c2], acc[c1][c2]);
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
            if (c < DH && cp < DH) Num<T>::st(dw + (k * DH + c) * DH + cp, acc[c1][c2][r]);
          }
    }
    // (3) dXtil[e][c] = sum_k t_k(lam_e) sum_c' dYtil[e][c'] W_k[c][c']   (overwrites xt)
#pragma unroll
    for (int et = 0; et < ET_MAX; ++et) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) xt[et][ct] = zero4();
      if (et < ET) {
        const int e = 16 * et + lq;
        float tk[kLMaxOrder];
        cheb_poly_lp(e < a.K ? lam[e] : 0.0f, a.P, tk);
#pragma unroll
        for (int k = 0; k < kLMaxOrder; ++k) {
          if (k < a.P) {
#pragma unroll
            for (int c2 = 0; c2 < CT; ++c2) {
              float dv4[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) dv4[r] = dytT[c2][et][r] * tk[k];
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                float wv[4];
                w_row4_t<T, DH>(w, k, ct, c2, lq, g, wv);
                xt[et][ct] = Num<T>::mma4(dv4, wv, xt[et][ct]);
              }
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
            f32x4 ub;
#pragma unroll
            for (int r = 0; r < 4; ++r) ub[r] = u_at(U, a.K, node, 16 * et + 4 * g + r, n);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dx[ct] = mma_acc<T>(ub, xt[et][ct], dx[ct]);
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) store_acc_t<T, DH>(a.dx, a.xsb, a.xsn, b, h, 16 * nt, a.N, ct, lq, g, dx[ct]);
    }
  }
  // dbias partial of this block
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    float s = dbias[ct][0] + dbias[ct][1] + dbias[ct][2] + dbias[ct][3];
    s += shfl_xor(s, 16);
    s += shfl_xor(s, 32);
    const int c = 16 * ct + lq;
    if (g == 0 && c < DH) a.dbias_part[(int64_t)item * DH + c] = s;
  }
}

// ---- launchers --------------------------------------------------------------------------------------------------
template <class T, int DH>
int launch_attn_fwd_lp(const AttnArgsT<T>& a, hipStream_t stream) {
  const dim3 grid((a.total + kLWaves - 1) / kLWaves), block(64 * kLWaves);
  const int kt = (a.N + 15) / 16;
#define FETA_LP_FWD(KT)                                                                         \
  {                                                                                             \
    const size_t lds = a.attn != nullptr ? sizeof(float) * kLWaves * 16 * (16 * KT + 1) : 0;   \
    auto kern = attn_fwd_lp_kernel<T, DH, KT>;                                                  \
    static LdsSeen seen;                                                                        \
    allow_dynamic_lds(kern, lds, seen);                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                      \
  }
  if (kt <= 4) FETA_LP_FWD(4) else if (kt <= 8) FETA_LP_FWD(8) else FETA_LP_FWD(16)
#undef FETA_LP_FWD
  return check_launch("feta_attn_fwd_bf16");
}

template <class T, int DH>
int launch_attn_bwd_lp(const AttnArgsT<T>& a, hipStream_t stream) {
  const dim3 grid((a.total + kLWaves - 1) / kLWaves), block(64 * kLWaves);
  auto k1 = attn_bwd_dq_lp_kernel<T, DH>;
  hipLaunchKernelGGL(k1, grid, block, 0, stream, a);
  int rc = check_launch("feta_attn_bwd_bf16 (dq)");
  if (rc != FETA_OK) return rc;
  auto k2 = attn_bwd_dkdv_lp_kernel<T, DH>;
  hipLaunchKernelGGL(k2, grid, block, 0, stream, a);
  return check_launch("feta_attn_bwd_bf16 (dk, dv)");
}

template <class T, int DH>
int launch_spec_lp(const FilterArgsT<T>& a, bool bwd, hipStream_t stream) {
  const dim3 grid((a.total + kLWaves - 1) / kLWaves), block(64 * kLWaves);
  const int et = (a.K + 15) / 16;
#define FETA_LP_SPEC(ET)                                              \
  {                                                                   \
    if (bwd) {                                                        \
      auto kern = spec_bwd_lp_kernel<T, DH, ET>;                      \
      hipLaunchKernelGGL(kern, grid, block, 0, stream, a);            \
    } else {                                                          \
      auto kern = spec_fwd_lp_kernel<T, DH, ET>;                      \
      hipLaunchKernelGGL(kern, grid, block, 0, stream, a);            \
    }                                                                 \
  }
  if (et <= 1) FETA_LP_SPEC(1) else if (et <= 2) FETA_LP_SPEC(2) else FETA_LP_SPEC(4)
#undef FETA_LP_SPEC
  return check_launch(bwd ? "feta_spec_filter_bwd_bf16" : "feta_spec_filter_fwd_bf16");
}

inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

}  // namespace feta

using namespace feta;

#define FETA_LP_DH_SWITCH(dh, CALL)                                      \
  switch (dh) {                                                          \
    case 16: return CALL(16);                                            \
    case 32: return CALL(32);                                            \
    case 64: return CALL(64);                                            \
    default: break;                                                      \
  }

extern "C" int feta_attn_fwd_bf16(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                  const void* pe, const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn,
                                  void* attn, float* stats, float scale, int B, int N, int H, int dh,
                                  feta_stream_t stream) {
  FETA_REQUIRE(q && k && v && n_real && out && stats, "attn_fwd_bf16: null pointer");
  FETA_REQUIRE(B > 0 && H > 0 && N >= 1 && N <= FETA_MAX_NODES, "attn_fwd_bf16: N=%d outside [1,%d]", N, FETA_MAX_NODES);
  FETA_REQUIRE(dh == 16 || dh == 32 || dh == 64, "attn_fwd_bf16: head dim %d not in {16,32,64}", dh);
  FETA_REQUIRE(aligned8(q) && aligned8(k) && aligned8(v) && aligned8(out) && (qkv_sb & 3) == 0 && (qkv_sn & 3) == 0 &&
               (o_sb & 3) == 0 && (o_sn & 3) == 0, "attn_fwd_bf16: token tensors must be 8-byte aligned, strides %% 4 == 0");
  AttnArgsT<bf16_t> a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.pe = (const bf16_t*)pe; a.n_real = n_real;
  a.out_w = (bf16_t*)out; a.attn = (bf16_t*)attn; a.stats = stats; a.qsb = qkv_sb; a.qsn = qkv_sn; a.osb = o_sb;
  a.osn = o_sn; a.scale = scale; a.B = B; a.N = N; a.H = H; a.NB = (N + 15) / 16; a.total = B * H * a.NB;
  a.drop_thresh = 0u; a.drop_scale = 1.0f;
#define CALL(D) launch_attn_fwd_lp<bf16_t, D>(a, (hipStream_t)stream)
  FETA_LP_DH_SWITCH(dh, CALL)
#undef CALL
  return FETA_E_ARG;
}

extern "C" int feta_attn_bwd_bf16(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                  const void* pe, const int32_t* n_real, const void* out, const void* dout,
                                  int64_t o_sb, int64_t o_sn, const float* stats, float* delta, void* dq, void* dk,
                                  void* dv, float scale, int B, int N, int H, int dh, feta_stream_t stream) {
  FETA_REQUIRE(q && k && v && n_real && out && dout && stats && delta && dq && dk && dv, "attn_bwd_bf16: null pointer");
  FETA_REQUIRE(B > 0 && H > 0 && N >= 1 && N <= FETA_MAX_NODES, "attn_bwd_bf16: N=%d outside [1,%d]", N, FETA_MAX_NODES);
  FETA_REQUIRE(dh == 16 || dh == 32 || dh == 64, "attn_bwd_bf16: head dim %d not in {16,32,64}", dh);
  FETA_REQUIRE(aligned8(q) && aligned8(k) && aligned8(v) && aligned8(out) && aligned8(dout) && (qkv_sb & 3) == 0 &&
               (qkv_sn & 3) == 0 && (o_sb & 3) == 0 && (o_sn & 3) == 0,
               "attn_bwd_bf16: token tensors must be 8-byte aligned, strides %% 4 == 0");
  AttnArgsT<bf16_t> a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.pe = (const bf16_t*)pe; a.n_real = n_real;
  a.out = (const bf16_t*)out; a.dout = (const bf16_t*)dout; a.stats_in = stats; a.delta = delta;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.qsb = qkv_sb; a.qsn = qkv_sn; a.osb = o_sb; a.osn = o_sn;
  a.scale = scale; a.B = B; a.N = N; a.H = H; a.NB = (N + 15) / 16; a.total = B * H * a.NB;
  a.drop_thresh = 0u; a.drop_scale = 1.0f;
#define CALL(D) launch_attn_bwd_lp<bf16_t, D>(a, (hipStream_t)stream)
  FETA_LP_DH_SWITCH(dh, CALL)
#undef CALL
  return FETA_E_ARG;
}

namespace {
template <class T>
void set_drop(AttnArgsT<T>& a, float p_drop, uint64_t seed, uint64_t offset, const uint64_t* dstate = nullptr,
              int clamp5 = 0) {
  a.dstate = reinterpret_cast<const unsigned long long*>(dstate);
  a.clamp5 = clamp5;
  if (p_drop > 0.0f) {
    const double t = (double)p_drop * 4294967296.0;
    a.drop_thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
    if (a.drop_thresh == 0u) a.drop_thresh = 1u;
    a.drop_scale = 1.0f / (1.0f - p_drop);
  } else {
    a.drop_thresh = 0u;
    a.drop_scale = 1.0f;
  }
  a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
  a.off_lo = (unsigned)offset; a.off_hi = (unsigned)(offset >> 32);
}

template <class T>
int attn_fwd_drop_t(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn, const void* pe,
                    const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn, void* attn, float* stats, float scale,
                    float p_drop, uint64_t seed, uint64_t offset, int B, int N, int H, int dh, hipStream_t stream,
                    const uint64_t* dstate = nullptr, int clamp5 = 0) {
  AttnArgsT<T> a{};
  a.q = (const T*)q; a.k = (const T*)k; a.v = (const T*)v; a.pe = (const T*)pe; a.n_real = n_real;
  a.out_w = (T*)out; a.attn = (T*)attn; a.stats = stats; a.qsb = qkv_sb; a.qsn = qkv_sn; a.osb = o_sb;
  a.osn = o_sn; a.scale = scale; a.B = B; a.N = N; a.H = H; a.NB = (N + 15) / 16; a.total = B * H * a.NB;
  set_drop(a, p_drop, seed, offset, dstate, clamp5);
  switch (dh) {
    case 16: return launch_attn_fwd_lp<T, 16>(a, stream);
    case 32: return launch_attn_fwd_lp<T, 32>(a, stream);
    case 64: return launch_attn_fwd_lp<T, 64>(a, stream);
    default: return FETA_E_ARG;
  }
}

template <class T>
int attn_bwd_drop_t(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn, const void* pe,
                    const int32_t* n_real, const void* out, const void* dout, int64_t o_sb, int64_t o_sn,
                    const float* stats, float* delta, void* dq, void* dk, void* dv, float scale, float p_drop,
                    uint64_t seed, uint64_t offset, int B, int N, int H, int dh, hipStream_t stream,
                    const uint64_t* dstate = nullptr, int clamp5 = 0) {
  AttnArgsT<T> a{};
  a.q = (const T*)q; a.k = (const T*)k; a.v = (const T*)v; a.pe = (const T*)pe; a.n_real = n_real;
  a.out = (const T*)out; a.dout = (const T*)dout; a.stats_in = stats; a.delta = delta;
  a.dq = (T*)dq; a.dk = (T*)dk; a.dv = (T*)dv; a.qsb = qkv_sb; a.qsn = qkv_sn; a.osb = o_sb; a.osn = o_sn;
  a.scale = scale; a.B = B; a.N = N; a.H = H; a.NB = (N + 15) / 16; a.total = B * H * a.NB;
  set_drop(a, p_drop, seed, offset, dstate, clamp5);
  switch (dh) {
    case 16: return launch_attn_bwd_lp<T, 16>(a, stream);
    case 32: return launch_attn_bwd_lp<T, 32>(a, stream);
    case 64: return launch_attn_bwd_lp<T, 64>(a, stream);
    default: return FETA_E_ARG;
  }
}
}  // namespace

static int attn_fwd_drop_any(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                  const void* pe, const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn,
                                  void* attn, float* stats, float scale, float p_drop, uint64_t seed, uint64_t offset,
                                  int dtype, int B, int N, int H, int dh, feta_stream_t stream, const uint64_t* dstate, int stab = 0) {
  FETA_REQUIRE(q && k && v && n_real && out && stats, "attn_fwd_drop: null pointer");
  FETA_REQUIRE(B > 0 && H > 0 && N >= 1 && N <= FETA_MAX_NODES, "attn_fwd_drop: N=%d outside [1,%d]", N, FETA_MAX_NODES);
  FETA_REQUIRE(dh == 16 || dh == 32 || dh == 64, "attn_fwd_drop: head dim %d not in {16,32,64}", dh);
  FETA_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "attn_fwd_drop: p = %g outside [0, 1)", (double)p_drop);
  FETA_REQUIRE(dtype == FETA_DTYPE_F32 || dtype == FETA_DTYPE_BF16, "attn_fwd_drop: unknown dtype %d", dtype);
  FETA_REQUIRE((qkv_sb & 3) == 0 && (qkv_sn & 3) == 0 && (o_sb & 3) == 0 && (o_sn & 3) == 0 &&
               (dtype == FETA_DTYPE_F32 ? (aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out))
                                        : (aligned8(q) && aligned8(k) && aligned8(v) && aligned8(out))),
               "attn_fwd_drop: misaligned token tensors / strides");
  if (dtype == FETA_DTYPE_F32)
    return attn_fwd_drop_t<float>(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, o_sb, o_sn, attn, stats, scale, p_drop, seed,
                                  offset, B, N, H, dh, (hipStream_t)stream, dstate, stab);
  return attn_fwd_drop_t<bf16_t>(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, o_sb, o_sn, attn, stats, scale, p_drop, seed,
                                 offset, B, N, H, dh, (hipStream_t)stream, dstate, stab);
}

extern "C" int feta_attn_fwd_drop(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                  const void* pe, const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn,
                                  void* attn, float* stats, float scale, float p_drop, uint64_t seed, uint64_t offset,
                                  int dtype, int B, int N, int H, int dh, feta_stream_t stream) {
  return attn_fwd_drop_any(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, o_sb, o_sn, attn, stats, scale, p_drop, seed, offset,
                           dtype, B, N, H, dh, stream, nullptr);
}

extern "C" int feta_attn_fwd_drop_dev(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                      const void* pe, const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn,
                                      void* attn, float* stats, float scale, float p_drop, const uint64_t* state,
                                      uint64_t offset_add, int dtype, int B, int N, int H, int dh, feta_stream_t stream) {
  FETA_REQUIRE(state != nullptr, "attn_fwd_drop_dev: null state");
  return attn_fwd_drop_any(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, o_sb, o_sn, attn, stats, scale, p_drop, 0, offset_add,
                           dtype, B, N, H, dh, stream, state);
}

static int attn_bwd_drop_any(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                  const void* pe, const int32_t* n_real, const void* out, const void* dout,
                                  int64_t o_sb, int64_t o_sn, const float* stats, float* delta, void* dq, void* dk,
                                  void* dv, float scale, float p_drop, uint64_t seed, uint64_t offset, int dtype,
                                  int B, int N, int H, int dh, feta_stream_t stream, const uint64_t* dstate, int stab = 0) {
  FETA_REQUIRE(q && k && v && n_real && out && dout && stats && delta && dq && dk && dv, "attn_bwd_drop: null pointer");
  FETA_REQUIRE(B > 0 && H > 0 && N >= 1 && N <= FETA_MAX_NODES, "attn_bwd_drop: N=%d outside [1,%d]", N, FETA_MAX_NODES);
  FETA_REQUIRE(dh == 16 || dh == 32 || dh == 64, "attn_bwd_drop: head dim %d not in {16,32,64}", dh);
  FETA_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "attn_bwd_drop: p = %g outside [0, 1)", (double)p_drop);
  FETA_REQUIRE(dtype == FETA_DTYPE_F32 || dtype == FETA_DTYPE_BF16, "attn_bwd_drop: unknown dtype %d", dtype);
  FETA_REQUIRE((qkv_sb & 3) == 0 && (qkv_sn & 3) == 0 && (o_sb & 3) == 0 && (o_sn & 3) == 0,
               "attn_bwd_drop: strides %% 4 == 0");
  if (dtype == FETA_DTYPE_F32)
    return attn_bwd_drop_t<float>(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, dout, o_sb, o_sn, stats, delta, dq, dk, dv,
                                  scale, p_drop, seed, offset, B, N, H, dh, (hipStream_t)stream, dstate, stab);
  return attn_bwd_drop_t<bf16_t>(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, dout, o_sb, o_sn, stats, delta, dq, dk, dv,
                                 scale, p_drop, seed, offset, B, N, H, dh, (hipStream_t)stream, dstate, stab);
}

extern "C" int feta_attn_bwd_drop(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                  const void* pe, const int32_t* n_real, const void* out, const void* dout,
                                  int64_t o_sb, int64_t o_sn, const float* stats, float* delta, void* dq, void* dk,
                                  void* dv, float scale, float p_drop, uint64_t seed, uint64_t offset, int dtype,
                                  int B, int N, int H, int dh, feta_stream_t stream) {
  return attn_bwd_drop_any(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, dout, o_sb, o_sn, stats, delta, dq, dk, dv, scale,
                           p_drop, seed, offset, dtype, B, N, H, dh, stream, nullptr);
}

extern "C" int feta_attn_bwd_drop_dev(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                      const void* pe, const int32_t* n_real, const void* out, const void* dout,
                                      int64_t o_sb, int64_t o_sn, const float* stats, float* delta, void* dq, void* dk,
                                      void* dv, float scale, float p_drop, const uint64_t* state, uint64_t offset_add,
                                      int dtype, int B, int N, int H, int dh, feta_stream_t stream) {
  FETA_REQUIRE(state != nullptr, "attn_bwd_drop_dev: null state");
  return attn_bwd_drop_any(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, dout, o_sb, o_sn, stats, delta, dq, dk, dv, scale,
                           p_drop, 0, offset_add, dtype, B, N, H, dh, stream, state);
}

extern "C" int feta_attn_fwd_stab(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                  const void* pe, const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn,
                                  void* attn, float* stats, float scale, int stab, int dtype, int B, int N, int H,
                                  int dh, feta_stream_t stream) {
  FETA_REQUIRE(stab == FETA_STAB_ROWMAX || stab == FETA_STAB_CLAMP5, "attn_fwd_stab: stab %d", stab);
  return attn_fwd_drop_any(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, o_sb, o_sn, attn, stats, scale, 0.0f, 0, 0, dtype, B,
                           N, H, dh, stream, nullptr, stab);
}

extern "C" int feta_attn_bwd_stab(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                                  const void* pe, const int32_t* n_real, const void* out, const void* dout,
                                  int64_t o_sb, int64_t o_sn, const float* stats, float* delta, void* dq, void* dk,
                                  void* dv, float scale, int stab, int dtype, int B, int N, int H, int dh,
                                  feta_stream_t stream) {
  FETA_REQUIRE(stab == FETA_STAB_ROWMAX || stab == FETA_STAB_CLAMP5, "attn_bwd_stab: stab %d", stab);
  return attn_bwd_drop_any(q, k, v, qkv_sb, qkv_sn, pe, n_real, out, dout, o_sb, o_sn, stats, delta, dq, dk, dv, scale, 0.0f,
                           0, 0, dtype, B, N, H, dh, stream, nullptr, stab);
}

static int spec_args_bf16(FilterArgsT<bf16_t>& a, const void* x, int64_t x_sb, int64_t x_sn, const void* u,
                          const float* lam, const void* coeff, const int32_t* n_real, int64_t y_sb, int64_t y_sn, int B,
                          int N, int H, int dh, int P, int K, int share) {
  FETA_REQUIRE(x && u && lam && coeff && n_real, "spec_filter_bf16: null pointer");
  FETA_REQUIRE(B > 0 && H > 0 && N >= 1 && N <= FETA_MAX_NODES, "spec_filter_bf16: N=%d outside [1,%d]", N, FETA_MAX_NODES);
  FETA_REQUIRE(dh == 16 || dh == 32, "spec_filter_bf16: head dim %d not in {16,32}", dh);
  FETA_REQUIRE(P >= 1 && P <= kLMaxOrder, "spec_filter_bf16: order %d outside [1,%d]", P, kLMaxOrder);
  FETA_REQUIRE(K >= 1 && K <= 64, "spec_filter_bf16: K=%d outside [1,64] (the bf16 path keeps the spectrum in registers)", K);
  FETA_REQUIRE(aligned8(x) && aligned8(coeff) && (x_sb & 3) == 0 && (x_sn & 3) == 0 && (y_sb & 3) == 0 && (y_sn & 3) == 0,
               "spec_filter_bf16: 8-byte aligned token tensors / weights, strides %% 4 == 0");
  a.x = (const bf16_t*)x; a.u = (const bf16_t*)u; a.lam = lam; a.coeff = (const bf16_t*)coeff; a.n_real = n_real;
  a.xsb = x_sb; a.xsn = x_sn; a.ysb = y_sb; a.ysn = y_sn; a.B = B; a.N = N; a.H = H; a.P = P; a.K = K; a.share = share;
  a.total = B * H;
  return FETA_OK;
}

extern "C" int feta_spec_filter_fwd_bf16(const void* x, int64_t x_sb, int64_t x_sn, const void* u, const float* lam,
                                         const void* coeff, const float* bias, const int32_t* n_real, void* y,
                                         int64_t y_sb, int64_t y_sn, int B, int N, int H, int dh, int P, int K,
                                         int heads_share_graph, feta_stream_t stream) {
  FilterArgsT<bf16_t> a{};
  int rc = spec_args_bf16(a, x, x_sb, x_sn, u, lam, coeff, n_real, y_sb, y_sn, B, N, H, dh, P, K, heads_share_graph);
  if (rc != FETA_OK) return rc;
  FETA_REQUIRE(y != nullptr && aligned8(y), "spec_filter_fwd_bf16: y");
  a.bias = bias; a.y = (bf16_t*)y;
  if (dh == 16) return launch_spec_lp<bf16_t, 16>(a, false, (hipStream_t)stream);
  return launch_spec_lp<bf16_t, 32>(a, false, (hipStream_t)stream);
}

extern "C" int feta_spec_filter_bwd_bf16(const void* x, int64_t x_sb, int64_t x_sn, const void* u, const float* lam,
                                         const void* coeff, const int32_t* n_real, const void* dy, int64_t y_sb,
                                         int64_t y_sn, void* dx, void* dcoeff, float* dbias_part, int B, int N, int H,
                                         int dh, int P, int K, int heads_share_graph, feta_stream_t stream) {
  FilterArgsT<bf16_t> a{};
  int rc = spec_args_bf16(a, x, x_sb, x_sn, u, lam, coeff, n_real, y_sb, y_sn, B, N, H, dh, P, K, heads_share_graph);
  if (rc != FETA_OK) return rc;
  FETA_REQUIRE(dy && dx && dcoeff && dbias_part && aligned8(dy) && aligned8(dx), "spec_filter_bwd_bf16: null / misaligned");
  a.dy = (const bf16_t*)dy; a.dx = (bf16_t*)dx; a.dcoeff = (bf16_t*)dcoeff; a.dbias_part = dbias_part;
  if (dh == 16) return launch_spec_lp<bf16_t, 16>(a, true, (hipStream_t)stream);
  return launch_spec_lp<bf16_t, 32>(a, true, (hipStream_t)stream);
}
