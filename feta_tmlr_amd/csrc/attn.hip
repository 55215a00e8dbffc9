// A1 attention core for gfx950: scores, masked exp, multiplicative positional kernel,
// clamped normalisation and the weighted sum, forward and backward, one wave per
// 16-row block.  Replaces the body of the (absent) DiffTransformerEncoderLayer.self_attn
// of the reference - contract: transformer/models.py:166-167,179,244,275; form
// witnesses: LSPE/layers/graphit_gt_layer.py:39-43,120-131,164.
//
// Work decomposition (graphs are small: N <= 256, dh <= 64):
//   forward / dq : one wave per (graph b, head h, 16-query block); the score tile is
//                  held TRANSPOSED (rows = keys on registers, column = query on the
//                  lane) so that softmax statistics are lane-local + two shuffles and
//                  the probabilities feed the P.V / dS.K MFMAs straight from the
//                  accumulator registers (contraction over keys = accumulator rows);
//   dk, dv       : one wave per (b, h, 16-key block) holding the tile un-transposed
//                  (rows = queries) so that dV = P^T dO and dK = dS^T Q contract over
//                  accumulator rows as well.  No atomics, no cross-wave reduction:
//                  results are bitwise reproducible.
#include <cmath>

#include "feta_abi_common.h"
#include <cstdlib>

#include "feta_tiles.h"

namespace feta {

struct AttnArgs {
  const float* q;
  const float* k;
  const float* v;
  const float* pe;
  const int32_t* n_real;
  const float* out;   // forward output (backward input)
  const float* dout;  // backward input
  const float* dout2; // optional second gradient into the same output (added to dout), dense path
  const float* stats_in;
  float* out_w;
  float* attn;
  float* stats;
  float* delta;
  float* dq;
  float* dk;
  float* dv;
  int64_t qsb, qsn, osb, osn;
  float scale;
  int B, N, H, NB;  // NB = ceil(N / 16) row blocks
  int total;        // B * H * NB work items
};

constexpr int kWaves = 4;

template <int DH, int KT_MAX>
__global__ __launch_bounds__(64 * kWaves) void attn_fwd_kernel(AttnArgs a) {
  constexpr int CT = Feat<DH>::CT;
  constexpr int KP = 16 * KT_MAX + 1;  // LDS row pitch of the staged probability block
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kWaves + wave_id();
  if (item >= a.total) return;
  const int qb = item % a.NB;
  const int bh = item / a.NB;
  const int h = bh % a.H, b = bh / a.H;
  const int n = a.n_real[b];
  const int q0 = 16 * qb;
  const int KT = (n + 15) >> 4;
  const int q = q0 + lq;  // this lane's query (column of the transposed tile)

  // every load below is UNCONDITIONAL (row indices clamped into the tensor, validity applied by selects):
  // a branch around a load serialises the memory round trips of the unrolled loops, without it the loads
  // of a phase go out back to back and the wave pays one latency per phase
  const int nm1 = max(n - 1, 0), qc = min(q, a.N - 1);
  Feat<DH> qf;
  load_row_sel<DH>(qf, tok_row(a.q, a.qsb, a.qsn, b, qc, h, DH), q < a.N, g, a.scale);

  // S^T tiles: acc[kt][r] <-> key 16kt + 4g + r, query q
  // (scalars, not f32x4 acc[KT_MAX]: the optimizer promotes such an array to ONE 4 KT_MAX-wide vector value and every
  // conditional tile update then copies the whole tuple - 256 + 127 registers at KT_MAX = 8; see csrc/attnout.hip)
  float acc[KT_MAX][4];
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    f32x4 t = zero4();
    if (kt < KT) {
      const int key = 16 * kt + lq;
      Feat<DH> kf;
      load_row_sel<DH>(kf, tok_row(a.k, a.qsb, a.qsn, b, min(key, nm1), h, DH), key < n, g);
      t = dot_rows<DH>(kf, qf, zero4());
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[kt][r] = t[r];
  }

  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    if (kt < KT) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * kt + 4 * g + r < n) m = fmaxf(m, acc[kt][r]);
    }
  }
  m = fmaxf(m, shfl_xor(m, 16));
  m = fmaxf(m, shfl_xor(m, 32));

  const bool has_pe = a.pe != nullptr;
  const float* pe_row = has_pe ? a.pe + ((int64_t)b * a.N + qc) * a.N : nullptr;
  float z = 0.0f;
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    if (kt < KT) {
      float pv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pv[r] = has_pe ? pe_row[min(16 * kt + 4 * g + r, nm1)] : 1.0f;
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

  // out = P . V : contraction over keys = accumulator rows
  f32x4 o[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) o[ct] = zero4();
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    if (kt < KT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[kt][r] *= rinv;
        const int key = 16 * kt + 4 * g + r;
        const float* vrow = tok_row(a.v, a.qsb, a.qsn, b, min(key, nm1), h, DH);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int c = 16 * ct + lq;
          const float vv = vrow[c < DH ? c : 0];
          o[ct] = mfma16(acc[kt][r], (key < n && c < DH) ? vv : 0.0f, o[ct]);
        }
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * ct + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 4 * g + r;
      if (qq < a.N && c < DH) tok_row(a.out_w, a.osb, a.osn, b, qq, h, DH)[c] = o[ct][r];
    }
  }

  // attn[b,h,q0:q0+16,:] is one contiguous run: stage the block in LDS, store coalesced
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
    float* dst = a.attn + ((int64_t)bh * a.N + q0) * a.N;
    int qq = 0, kk = lane;
    while (kk >= a.N) {
      kk -= a.N;
      ++qq;
    }
    for (int idx = lane; idx < rows * a.N; idx += 64) {
      dst[idx] = kk < cols ? st[qq * KP + kk] : 0.0f;
      kk += 64;
      while (kk >= a.N) {
        kk -= a.N;
        ++qq;
      }
    }
  }
}

// dq for one 16-query block (transposed tile, as the forward) + delta = rowsum(dout*out)
template <int DH>
__global__ __launch_bounds__(64 * kWaves) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int CT = Feat<DH>::CT;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kWaves + wave_id();
  if (item >= a.total) return;
  const int qb = item % a.NB;
  const int bh = item / a.NB;
  const int h = bh % a.H, b = bh / a.H;
  const int n = a.n_real[b];
  const int q0 = 16 * qb;
  const int KT = (n + 15) >> 4;
  const int q = q0 + lq;
  const bool qok = q < a.N;

  Feat<DH> qf, dof, of;
  load_row<DH>(qf, qok ? tok_row(a.q, a.qsb, a.qsn, b, q, h, DH) : nullptr, g, a.scale);
  load_row<DH>(dof, qok ? tok_row(a.dout, a.osb, a.osn, b, q, h, DH) : nullptr, g);
  load_row<DH>(of, qok ? tok_row(a.out, a.osb, a.osn, b, q, h, DH) : nullptr, g);
  float delta = 0.0f;
#pragma unroll
  for (int j = 0; j < Feat<DH>::NJ; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) delta += dof.f[j][s] * of.f[j][s];
  delta += shfl_xor(delta, 16);
  delta += shfl_xor(delta, 32);
  if (g == 0 && qok) a.delta[(int64_t)bh * a.N + q] = delta;

  float m = 0.0f, z = 1.0f;
  if (qok) {
    const float* st = a.stats_in + ((int64_t)bh * a.N + q) * 2;
    m = st[0];
    z = st[1];
  }
  const float rinv = 1.0f / fmaxf(z, 1e-6f);
  if (z < 1e-6f) delta = 0.0f;  // clamp active: the normaliser is a constant
  f32x4 dq[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) dq[ct] = zero4();

  // one batch of unconditional, clamped loads per key tile (K and V rows in both operand layouts, pe);
  // the batch of tile kt + 1 is requested before tile kt is computed
  struct Tile {
    Feat<DH> kf, vf;
    float pv[4], kb[4][CT];
  };
  const int nm1 = max(n - 1, 0);
  const bool has_pe = a.pe != nullptr;
  const float* pe_c = has_pe ? a.pe + ((int64_t)b * a.N + min(q, a.N - 1)) * a.N : nullptr;
  auto load_tile = [&](int kt, Tile& T) {
    const int krow = 16 * kt + lq;
    load_row_sel<DH>(T.kf, tok_row(a.k, a.qsb, a.qsn, b, min(krow, nm1), h, DH), krow < n, g);
    load_row_sel<DH>(T.vf, tok_row(a.v, a.qsb, a.qsn, b, min(krow, nm1), h, DH), krow < n, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * g + r;
      const int kc = min(key, nm1);
      T.pv[r] = has_pe ? pe_c[kc] : 1.0f;
      const float* kr = tok_row(a.k, a.qsb, a.qsn, b, kc, h, DH);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = 16 * ct + lq;
        const float kv = kr[c < DH ? c : 0];
        T.kb[r][ct] = (key < n && c < DH) ? kv : 0.0f;
      }
    }
  };
  Tile cur;
  if (KT > 0) load_tile(0, cur);
  for (int kt = 0; kt < KT; ++kt) {
    Tile nxt;
    load_tile(min(kt + 1, KT - 1), nxt);
    f32x4 s = dot_rows<DH>(cur.kf, qf, zero4());    // scores^T
    f32x4 da = dot_rows<DH>(cur.vf, dof, zero4());  // (dout . v^T)^T
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * g + r;
      const float p = (key < n && qok) ? fast_exp(s[r] - m) * cur.pv[r] * rinv : 0.0f;
      const float ds = p * (da[r] - delta);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) dq[ct] = mfma16(ds, cur.kb[r][ct], dq[ct]);
    }
    cur = nxt;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * ct + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 4 * g + r;
      if (qq < a.N && c < DH) tok_row(a.dq, a.qsb, a.qsn, b, qq, h, DH)[c] = dq[ct][r] * a.scale;
    }
  }
}

// dk, dv for one 16-key block (un-transposed tile: rows = queries, column = key)
template <int DH>
__global__ __launch_bounds__(64 * kWaves) void attn_bwd_dkdv_kernel(AttnArgs a) {
  constexpr int CT = Feat<DH>::CT;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kWaves + wave_id();
  if (item >= a.total) return;
  const int kb = item % a.NB;
  const int bh = item / a.NB;
  const int h = bh % a.H, b = bh / a.H;
  const int n = a.n_real[b];
  const int key = 16 * kb + lq;  // this lane's key (column)
  const bool kok = key < n;

  f32x4 dk[CT], dv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    dk[ct] = zero4();
    dv[ct] = zero4();
  }

  if (16 * kb < n) {
    Feat<DH> kf, vf;
    load_row<DH>(kf, kok ? tok_row(a.k, a.qsb, a.qsn, b, key, h, DH) : nullptr, g);
    load_row<DH>(vf, kok ? tok_row(a.v, a.qsb, a.qsn, b, key, h, DH) : nullptr, g);
    // one batch of unconditional, clamped loads per query tile; the next tile's batch is requested first
    struct Tile {
      Feat<DH> qf, dof;
      float sm[4], sz[4], sd[4], pv[4], dob[4][CT], qbv[4][CT];
    };
    const int keyc = min(key, max(n - 1, 0));
    const bool has_pe = a.pe != nullptr;
    auto load_tile = [&](int qb, Tile& T) {
      const int qrow = 16 * qb + lq, qrc = min(qrow, a.N - 1);
      load_row_sel<DH>(T.qf, tok_row(a.q, a.qsb, a.qsn, b, qrc, h, DH), qrow < a.N, g, a.scale);
      load_row_sel<DH>(T.dof, tok_row(a.dout, a.osb, a.osn, b, qrc, h, DH), qrow < a.N, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = 16 * qb + 4 * g + r, qqc = min(qq, a.N - 1);
        const float* st = a.stats_in + ((int64_t)bh * a.N + qqc) * 2;
        T.sm[r] = st[0];
        T.sz[r] = st[1];
        T.sd[r] = a.delta[(int64_t)bh * a.N + qqc];
        T.pv[r] = has_pe ? a.pe[((int64_t)b * a.N + qqc) * a.N + keyc] : 1.0f;
        const float* dorow = tok_row(a.dout, a.osb, a.osn, b, qqc, h, DH);
        const float* qrow_p = tok_row(a.q, a.qsb, a.qsn, b, qqc, h, DH);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int c = 16 * ct + lq, cc = c < DH ? c : 0;
          const bool ok = qq < a.N && c < DH;
          const float dv_ = dorow[cc], qv_ = qrow_p[cc];
          T.dob[r][ct] = ok ? dv_ : 0.0f;
          T.qbv[r][ct] = ok ? qv_ * a.scale : 0.0f;
        }
      }
    };
    Tile cur;
    load_tile(0, cur);
    for (int qb = 0; qb < a.NB; ++qb) {
      Tile nxt;
      load_tile(min(qb + 1, a.NB - 1), nxt);
      f32x4 s = dot_rows<DH>(cur.qf, kf, zero4());    // scores: row = query 4g+r, col = key
      f32x4 da = dot_rows<DH>(cur.dof, vf, zero4());  // dout . v^T
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = 16 * qb + 4 * g + r;
        const float zz = cur.sz[r];
        const float p = (qq < a.N && kok) ? fast_exp(s[r] - cur.sm[r]) * cur.pv[r] * (1.0f / fmaxf(zz, 1e-6f)) : 0.0f;
        const float ds = p * (da[r] - (zz < 1e-6f ? 0.0f : cur.sd[r]));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          dv[ct] = mfma16(p, cur.dob[r][ct], dv[ct]);
          dk[ct] = mfma16(ds, cur.qbv[r][ct], dk[ct]);
        }
      }
      cur = nxt;
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * ct + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = 16 * kb + 4 * g + r;
      if (kk < a.N && c < DH) {
        tok_row(a.dk, a.qsb, a.qsn, b, kk, h, DH)[c] = dk[ct][r];
        tok_row(a.dv, a.qsb, a.qsn, b, kk, h, DH)[c] = dv[ct][r];
      }
    }
  }
}

// ---- small graphs (N <= 64): every global load is issued before the first MFMA ----------------
// The kernels above guard each load with a branch (rows beyond n_real do not exist for the
// maths), which serialises ~30 memory round trips per wave.  At ZINC/MUTAG sizes a wave's whole
// working set is a few dozen registers, so here every row index is CLAMPED into the tensor, the
// loads are unconditional and issued as one batch, and validity is applied by selects.

template <int DH, int KT_MAX>
__global__ __launch_bounds__(64 * kWaves) void attn_fwd_dense_kernel(AttnArgs a) {
  constexpr int CT = Feat<DH>::CT;
  constexpr int KP = 16 * KT_MAX + 1;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kWaves + wave_id();
  if (item >= a.total) return;
  const int qb = item % a.NB;
  const int bh = item / a.NB;
  const int h = bh % a.H, b = bh / a.H;
  const int n = a.n_real[b];
  const int q0 = 16 * qb;
  const int q = q0 + lq;
  const int qc = min(q, a.N - 1);
  const bool has_pe = a.pe != nullptr;

  // ---- load batch: q row, K rows of every key tile, pe and V elements -------------------------
  Feat<DH> qf, kf[KT_MAX];
  load_row_sel<DH>(qf, tok_row(a.q, a.qsb, a.qsn, b, qc, h, DH), q < a.N, g, a.scale);
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    const int key = 16 * kt + lq;
    load_row_sel<DH>(kf[kt], tok_row(a.k, a.qsb, a.qsn, b, min(key, n - 1), h, DH), key < n, g);
  }
  float pv[KT_MAX][4], vb[KT_MAX][4][CT];
  const float* pe_row = has_pe ? a.pe + ((int64_t)b * a.N + qc) * a.N : nullptr;
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * g + r;
      const int kc = min(key, n - 1);
      pv[kt][r] = has_pe ? pe_row[kc] : 1.0f;
      const float* vrow = tok_row(a.v, a.qsb, a.qsn, b, kc, h, DH);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = 16 * ct + lq;
        const float vv = vrow[c < DH ? c : 0];
        vb[kt][r][ct] = (key < n && c < DH) ? vv : 0.0f;
      }
    }
  }

  // ---- S^T tiles, softmax statistics ----------------------------------------------------------
  f32x4 acc[KT_MAX];
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) acc[kt] = dot_rows<DH>(kf[kt], qf, zero4());
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * kt + 4 * g + r < n) m = fmaxf(m, acc[kt][r]);
  m = fmaxf(m, shfl_xor(m, 16));
  m = fmaxf(m, shfl_xor(m, 32));
  float z = 0.0f;
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool kok = 16 * kt + 4 * g + r < n;
      const float e = kok ? fast_exp(acc[kt][r] - m) * pv[kt][r] : 0.0f;
      acc[kt][r] = e;
      z += e;
    }
  z += shfl_xor(z, 16);
  z += shfl_xor(z, 32);
  const float rinv = 1.0f / fmaxf(z, 1e-6f);
  if (g == 0 && q < a.N) {
    float* st = a.stats + ((int64_t)bh * a.N + q) * 2;
    st[0] = m;
    st[1] = z;
  }

  // ---- out = P . V ---------------------------------------------------------------------------
  f32x4 o[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) o[ct] = zero4();
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[kt][r] *= rinv;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) o[ct] = mfma16(acc[kt][r], vb[kt][r][ct], o[ct]);
    }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * ct + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 4 * g + r;
      if (qq < a.N && c < DH) tok_row(a.out_w, a.osb, a.osn, b, qq, h, DH)[c] = o[ct][r];
    }
  }

  if (a.attn != nullptr) {
    float* st = feta_lds + wave_id() * 16 * KP;
#pragma unroll
    for (int kt = 0; kt < KT_MAX; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[lq * KP + 16 * kt + 4 * g + r] = acc[kt][r];
    wave_lds_sync();
    const int rows = min(16, a.N - q0);
    float* dst = a.attn + ((int64_t)bh * a.N + q0) * a.N;
    int qq = 0, kk = lane;
    while (kk >= a.N) {
      kk -= a.N;
      ++qq;
    }
    for (int idx = lane; idx < rows * a.N; idx += 64) {
      dst[idx] = st[qq * KP + kk];   // columns >= n hold exact zeros
      kk += 64;
      while (kk >= a.N) {
        kk -= a.N;
        ++qq;
      }
    }
  }
}

template <int DH, int KT_MAX>
__device__ void attn_bwd_dq_dense_role(const AttnArgs& a, int item) {
  constexpr int CT = Feat<DH>::CT;
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  if (item >= a.total) return;
  const int qb = item % a.NB;
  const int bh = item / a.NB;
  const int h = bh % a.H, b = bh / a.H;
  const int n = a.n_real[b];
  const int q0 = 16 * qb;
  const int q = q0 + lq;
  const int qc = min(q, a.N - 1);
  const bool qok = q < a.N;
  const bool has_pe = a.pe != nullptr;

  Feat<DH> qf, dof, of, kf[KT_MAX], vf[KT_MAX];
  load_row_sel<DH>(qf, tok_row(a.q, a.qsb, a.qsn, b, qc, h, DH), qok, g, a.scale);
  load_row_sel<DH>(dof, tok_row(a.dout, a.osb, a.osn, b, qc, h, DH), qok, g);
  if (a.dout2 != nullptr) {
    Feat<DH> d2;
    load_row_sel<DH>(d2, tok_row(a.dout2, a.osb, a.osn, b, qc, h, DH), qok, g);
#pragma unroll
    for (int j = 0; j < Feat<DH>::NJ; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) dof.f[j][s] += d2.f[j][s];
  }
  load_row_sel<DH>(of, tok_row(a.out, a.osb, a.osn, b, qc, h, DH), qok, g);
  const float* st = a.stats_in + ((int64_t)bh * a.N + qc) * 2;
  const float m = st[0], z = st[1];
  float pv[KT_MAX][4], kb[KT_MAX][4][CT];
  const float* pe_row = has_pe ? a.pe + ((int64_t)b * a.N + qc) * a.N : nullptr;
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    const int krow = 16 * kt + lq;
    const int krc = min(krow, n - 1);
    load_row_sel<DH>(kf[kt], tok_row(a.k, a.qsb, a.qsn, b, krc, h, DH), krow < n, g);
    load_row_sel<DH>(vf[kt], tok_row(a.v, a.qsb, a.qsn, b, krc, h, DH), krow < n, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * kt + 4 * g + r;
      const int kc = min(key, n - 1);
      pv[kt][r] = has_pe ? pe_row[kc] : 1.0f;
      const float* kr = tok_row(a.k, a.qsb, a.qsn, b, kc, h, DH);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = 16 * ct + lq;
        const float kv = kr[c < DH ? c : 0];
        kb[kt][r][ct] = (key < n && c < DH) ? kv : 0.0f;
      }
    }
  }

  float delta = 0.0f;
#pragma unroll
  for (int j = 0; j < Feat<DH>::NJ; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) delta += dof.f[j][s] * of.f[j][s];
  delta += shfl_xor(delta, 16);
  delta += shfl_xor(delta, 32);
  if (g == 0 && qok) a.delta[(int64_t)bh * a.N + q] = delta;
  const float rinv = 1.0f / fmaxf(z, 1e-6f);
  if (z < 1e-6f) delta = 0.0f;

  f32x4 dq[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) dq[ct] = zero4();
#pragma unroll
  for (int kt = 0; kt < KT_MAX; ++kt) {
    const f32x4 s = dot_rows<DH>(kf[kt], qf, zero4());
    const f32x4 da = dot_rows<DH>(vf[kt], dof, zero4());
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool kok = 16 * kt + 4 * g + r < n;
      const float p = kok ? fast_exp(s[r] - m) * pv[kt][r] * rinv : 0.0f;
      const float ds = p * (da[r] - delta);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) dq[ct] = mfma16(ds, kb[kt][r][ct], dq[ct]);
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * ct + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 4 * g + r;
      if (qq < a.N && c < DH) tok_row(a.dq, a.qsb, a.qsn, b, qq, h, DH)[c] = dq[ct][r] * a.scale;
    }
  }
}

template <int DH, int KT_MAX>
__device__ void attn_bwd_dkdv_dense_role(const AttnArgs& a, int item) {
  constexpr int CT = Feat<DH>::CT;
  static_assert(CT == 1, "dense backward: one column tile (delta is a 16-lane row sum)");
  const int lane = lane_id(), lq = lane & 15, g = lane >> 4;
  if (item >= a.total) return;
  const int kblk = item % a.NB;
  const int bh = item / a.NB;
  const int h = bh % a.H, b = bh / a.H;
  const int n = a.n_real[b];
  const int key = 16 * kblk + lq;
  const bool kok = key < n;
  const int keyc = min(key, n - 1);
  const bool has_pe = a.pe != nullptr;

  // ---- load batch over every query block ------------------------------------------------------
  Feat<DH> kf, vf, qf[KT_MAX], dof[KT_MAX];
  load_row_sel<DH>(kf, tok_row(a.k, a.qsb, a.qsn, b, keyc, h, DH), kok, g);
  load_row_sel<DH>(vf, tok_row(a.v, a.qsb, a.qsn, b, keyc, h, DH), kok, g);
  float sm[KT_MAX][4], sz[KT_MAX][4], sd[KT_MAX][4], pv[KT_MAX][4];
  float dob[KT_MAX][4][CT], qbv[KT_MAX][4][CT];
#pragma unroll
  for (int qb = 0; qb < KT_MAX; ++qb) {
    const int qrow = 16 * qb + lq;
    const int qrc = min(qrow, a.N - 1);
    load_row_sel<DH>(qf[qb], tok_row(a.q, a.qsb, a.qsn, b, qrc, h, DH), qrow < a.N, g, a.scale);
    load_row_sel<DH>(dof[qb], tok_row(a.dout, a.osb, a.osn, b, qrc, h, DH), qrow < a.N, g);
    if (a.dout2 != nullptr) {
      Feat<DH> d2;
      load_row_sel<DH>(d2, tok_row(a.dout2, a.osb, a.osn, b, qrc, h, DH), qrow < a.N, g);
#pragma unroll
      for (int s = 0; s < 4; ++s) dof[qb].f[0][s] += d2.f[0][s];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = 16 * qb + 4 * g + r;
      const int qqc = min(qq, a.N - 1);
      const float* st = a.stats_in + ((int64_t)bh * a.N + qqc) * 2;
      sm[qb][r] = st[0];
      sz[qb][r] = st[1];
      pv[qb][r] = has_pe ? a.pe[((int64_t)b * a.N + qqc) * a.N + keyc] : 1.0f;
      const int cc = lq < DH ? lq : 0;
      const bool ok = qq < a.N && lq < DH;
      float dv_ = tok_row(a.dout, a.osb, a.osn, b, qqc, h, DH)[cc];
      if (a.dout2 != nullptr) dv_ += tok_row(a.dout2, a.osb, a.osn, b, qqc, h, DH)[cc];
      const float qv_ = tok_row(a.q, a.qsb, a.qsn, b, qqc, h, DH)[cc];
      const float ov_ = tok_row(a.out, a.osb, a.osn, b, qqc, h, DH)[cc];
      dob[qb][r][0] = ok ? dv_ : 0.0f;
      qbv[qb][r][0] = ok ? qv_ * a.scale : 0.0f;
      sd[qb][r] = ok ? dv_ * ov_ : 0.0f;   // summed over the 16 feature lanes below
    }
  }
  // delta[q] = rowsum(dout * out): recomputed here (16-lane DPP sum) so that this role does not
  // depend on the dq role and both run in ONE launch
#pragma unroll
  for (int qb = 0; qb < KT_MAX; ++qb)
#pragma unroll
    for (int r = 0; r < 4; ++r) sd[qb][r] = row16_sum(sd[qb][r]);

  f32x4 dk[CT], dv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    dk[ct] = zero4();
    dv[ct] = zero4();
  }
#pragma unroll
  for (int qb = 0; qb < KT_MAX; ++qb) {
    const f32x4 s = dot_rows<DH>(qf[qb], kf, zero4());
    const f32x4 da = dot_rows<DH>(dof[qb], vf, zero4());
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = (16 * qb + 4 * g + r < a.N) && kok;
      const float zz = sz[qb][r];
      const float p = ok ? fast_exp(s[r] - sm[qb][r]) * pv[qb][r] * (1.0f / fmaxf(zz, 1e-6f)) : 0.0f;
      const float ds = p * (da[r] - (zz < 1e-6f ? 0.0f : sd[qb][r]));
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        dv[ct] = mfma16(p, dob[qb][r][ct], dv[ct]);
        dk[ct] = mfma16(ds, qbv[qb][r][ct], dk[ct]);
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = 16 * ct + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = 16 * kblk + 4 * g + r;
      if (kk < a.N && c < DH) {
        tok_row(a.dk, a.qsb, a.qsn, b, kk, h, DH)[c] = dk[ct][r];
        tok_row(a.dv, a.qsb, a.qsn, b, kk, h, DH)[c] = dv[ct][r];
      }
    }
  }
}

template <int DH, int KT_MAX>
void launch_fwd_dense_t(const AttnArgs& a, hipStream_t stream) {
  const dim3 grid((a.total + kWaves - 1) / kWaves), block(64 * kWaves);
  const size_t lds = a.attn != nullptr ? sizeof(float) * kWaves * 16 * (16 * KT_MAX + 1) : 0;
  auto kern = attn_fwd_dense_kernel<DH, KT_MAX>;
  hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
}

// one launch: workgroups [0, nb) compute dq, [nb, 2 nb) compute dk and dv
template <int DH, int KT_MAX>
__global__ __launch_bounds__(64 * kWaves) void attn_bwd_dense_kernel(AttnArgs a, int nb) {
  if ((int)blockIdx.x < nb)
    attn_bwd_dq_dense_role<DH, KT_MAX>(a, blockIdx.x * kWaves + wave_id());
  else
    attn_bwd_dkdv_dense_role<DH, KT_MAX>(a, (blockIdx.x - nb) * kWaves + wave_id());
}

template <int DH, int KT_MAX>
void launch_bwd_dense_t(const AttnArgs& a, hipStream_t stream) {
  const int nb = (a.total + kWaves - 1) / kWaves;
  auto kern = attn_bwd_dense_kernel<DH, KT_MAX>;
  hipLaunchKernelGGL(kern, dim3(2 * nb), dim3(64 * kWaves), 0, stream, a, nb);
}

template <int DH, int KT_MAX>
void launch_fwd_t(const AttnArgs& a, hipStream_t stream) {
  const dim3 grid((a.total + kWaves - 1) / kWaves), block(64 * kWaves);
  // per-wave staging block of the probability tile: 16 rows x (16 KT_MAX + 1) floats
  const size_t lds = a.attn != nullptr ? sizeof(float) * kWaves * 16 * (16 * KT_MAX + 1) : 0;
  auto kern = attn_fwd_kernel<DH, KT_MAX>;
  static LdsSeen lds_seen;  // KT_MAX = 16: 65.8 KB, above the 64 KB default dynamic-LDS cap
  allow_dynamic_lds(kern, lds, lds_seen);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
}

template <int DH>
int launch_fwd(const AttnArgs& a, int kt_max, hipStream_t stream) {
  if constexpr (DH <= 16) {  // batched-load kernels: the register budget allows them up to N = 64
    if (kt_max <= 3) { launch_fwd_dense_t<DH, 3>(a, stream); return check_launch("feta_attn_fwd"); }
    if (kt_max <= 4) { launch_fwd_dense_t<DH, 4>(a, stream); return check_launch("feta_attn_fwd"); }
  }
  if (kt_max <= 3) launch_fwd_t<DH, 3>(a, stream);
  else if (kt_max <= 4) launch_fwd_t<DH, 4>(a, stream);
  else if (kt_max <= 8) launch_fwd_t<DH, 8>(a, stream);
  else launch_fwd_t<DH, 16>(a, stream);
  return check_launch("feta_attn_fwd");
}

// ---- backward, one workgroup per graph (4 heads x dh 16, N <= 64) -------------------------------
// 8 waves: wave = (head, role); role 0 computes dq over the head's query tiles, role 1 dk / dv over its
// key tiles (the arithmetic of the dense roles above).  What the per-(head, tile) kernels fetch again
// and again - every key tile re-reads q / dout / stats of all queries, every query tile K and V of all
// keys, 64 bytes at a time - is fetched ONCE per graph here: the 256-byte node rows of q, k, v, out and
// dout (+ dout2), pe_b and the statistics go to LDS with 16-byte requests, both roles take their
// operands (either layout) from there, and dq | dk | dv leave through the same tiles as whole rows.
constexpr int kGbP = 64 + 4;   // pitch of a staged 64-float row

__host__ __device__ inline int attn_bwd_graph_lds_floats(int nt) {
  const int nr = 16 * nt;
  return 5 * nr * kGbP + nr * (nr + 1) + 4 * nr * 2;
}

template <int NT>
__global__ __launch_bounds__(512) void attn_bwd_graph_kernel(AttnArgs a) {
  constexpr int DH = 16, H = 4, P = kGbP, NR = 16 * NT, PEP = NR + 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lq = lane & 15, g = lane >> 4;
  const int h = wv & 3, role = wv >> 2;
  const int b = blockIdx.x;
  const int n = a.n_real[b];
  float* Qs = feta_lds;        // [NR][P] q, later dq
  float* Ks = Qs + NR * P;     // k, later dk
  float* Vs = Ks + NR * P;     // v, later dv
  float* Ds = Vs + NR * P;     // dout (+ dout2)
  float* Os = Ds + NR * P;     // out
  float* PE = Os + NR * P;     // [NR][PEP]
  float* ST = PE + NR * PEP;   // [H][NR][2]
  const bool has_pe = a.pe != nullptr;
  const int nm1 = a.N - 1;

  // ---- cooperative loads: NT * 16 * 16 float4 per tensor, 512 threads ------------------------------------
  constexpr int RI = (NR * 16 + 511) / 512;
  float4 qv[RI], kv[RI], vv[RI], dv_[RI], ov[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int idx = min(tid + 512 * i, NR * 16 - 1), node = min(idx >> 4, nm1), c4 = 4 * (idx & 15);
    qv[i] = *reinterpret_cast<const float4*>(tok_row(a.q, a.qsb, a.qsn, b, node, 0, DH) + c4);
    kv[i] = *reinterpret_cast<const float4*>(tok_row(a.k, a.qsb, a.qsn, b, node, 0, DH) + c4);
    vv[i] = *reinterpret_cast<const float4*>(tok_row(a.v, a.qsb, a.qsn, b, node, 0, DH) + c4);
    ov[i] = *reinterpret_cast<const float4*>(tok_row(a.out, a.osb, a.osn, b, node, 0, DH) + c4);
    float4 d1 = *reinterpret_cast<const float4*>(tok_row(a.dout, a.osb, a.osn, b, node, 0, DH) + c4);
    if (a.dout2 != nullptr) {
      const float4 d2 = *reinterpret_cast<const float4*>(tok_row(a.dout2, a.osb, a.osn, b, node, 0, DH) + c4);
      d1 = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
    }
    dv_[i] = d1;
  }
  constexpr int PEI = (NR * NR + 511) / 512;
  float pev[PEI];
#pragma unroll
  for (int i = 0; i < PEI; ++i) {
    const int idx = tid + 512 * i, qq = idx / NR, kk = idx - qq * NR;
    const float v = has_pe ? a.pe[((int64_t)b * a.N + min(qq, nm1)) * a.N + min(kk, nm1)] : 1.0f;
    pev[i] = (idx < NR * NR && qq < a.N && kk < a.N) ? v : 0.0f;
  }
  {
    const int hh = tid / (NR * 2), rem = tid - hh * NR * 2;   // H * NR * 2 <= 512
    const float sv = a.stats_in[(((int64_t)b * H + min(hh, H - 1)) * a.N + min(rem >> 1, nm1)) * 2 + (rem & 1)];
    if (tid < H * NR * 2) ST[tid] = sv;
  }
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int idx = tid + 512 * i;
    if (idx < NR * 16) {
      const int off = (idx >> 4) * P + 4 * (idx & 15);
      *reinterpret_cast<float4*>(Qs + off) = qv[i];
      *reinterpret_cast<float4*>(Ks + off) = kv[i];
      *reinterpret_cast<float4*>(Vs + off) = vv[i];
      *reinterpret_cast<float4*>(Ds + off) = dv_[i];
      *reinterpret_cast<float4*>(Os + off) = ov[i];
    }
  }
#pragma unroll
  for (int i = 0; i < PEI; ++i) {
    const int idx = tid + 512 * i;
    if (idx < NR * NR) PE[(idx / NR) * PEP + idx % NR] = pev[i];
  }
  __syncthreads();

  const int co = DH * h;
  const int bh = b * H + h;
  f32x4 r0[NT], r1[NT];   // role 0: dq tiles; role 1: dk, dv tiles
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    r0[t] = zero4();
    r1[t] = zero4();
  }
  if (role == 0) {
    // dq: S^T orientation (key 4g+r, query lq)
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
      if (g == 0 && qok) a.delta[(int64_t)bh * a.N + q] = delta;
      const float m = ST[(h * NR + q) * 2], z = ST[(h * NR + q) * 2 + 1];
      const float rinv = 1.0f / fmaxf(z, 1e-6f);
      if (z < 1e-6f) delta = 0.0f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (16 * kt >= n) continue;
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
    // dk, dv: S orientation (query 4g+r, key lq)
    Feat<DH> qf[NT], dof[NT];
    float qb4[NT][4], dob[NT][4], sd[NT][4];
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
        const f32x4 s = dot_rows<DH>(qf[qb], kf, zero4());
        const f32x4 da = dot_rows<DH>(dof[qb], vf, zero4());
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * qb + 4 * g + r;
          const float m = ST[(h * NR + q) * 2], z = ST[(h * NR + q) * 2 + 1];
          const bool ok = q < a.N && key < n;
          const float p = ok ? fast_exp(s[r] - m) * PE[q * PEP + key] * (1.0f / fmaxf(z, 1e-6f)) : 0.0f;
          const float ds = p * (da[r] - (z < 1e-6f ? 0.0f : sd[qb][r]));
          r1[kt] = mfma16(p, dob[qb][r], r1[kt]);    // dv (key 4g+r, c lq)
          r0[kt] = mfma16(ds, qb4[qb][r], r0[kt]);   // dk
        }
      }
    }
  }
  __syncthreads();   // every wave has taken its operands: the q / k / v tiles become dq / dk / dv
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * t + 4 * g + r;
      if (role == 0) {
        Qs[rr * P + co + lq] = r0[t][r] * a.scale;
      } else {
        Ks[rr * P + co + lq] = r0[t][r];
        Vs[rr * P + co + lq] = r1[t][r];
      }
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int idx = tid + 512 * i, node = idx >> 4, c4 = 4 * (idx & 15);
    if (idx < NR * 16 && node < a.N) {
      const int off = node * P + c4;
      *reinterpret_cast<float4*>(tok_row(a.dq, a.qsb, a.qsn, b, node, 0, DH) + c4) = *reinterpret_cast<const float4*>(Qs + off);
      *reinterpret_cast<float4*>(tok_row(a.dk, a.qsb, a.qsn, b, node, 0, DH) + c4) = *reinterpret_cast<const float4*>(Ks + off);
      *reinterpret_cast<float4*>(tok_row(a.dv, a.qsb, a.qsn, b, node, 0, DH) + c4) = *reinterpret_cast<const float4*>(Vs + off);
    }
  }
}

template <int NT>
void launch_bwd_graph_t(const AttnArgs& a, hipStream_t stream) {
  const size_t lds = sizeof(float) * attn_bwd_graph_lds_floats(NT);
  auto kern = attn_bwd_graph_kernel<NT>;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(512), lds, stream, a);
}

// ---- backward, one workgroup per (graph, head): 4 heads x dh 16, 64 < N <= 256 (config 4, PATTERN) -----------------
// The per-(head, tile) kernels above issue 10 (dq) / 26 (dk, dv) gather instructions per tile pair and lane, every one
// of them 64 scattered 4-byte accesses: at N = 128 they are bound by the address unit (16 + 18 us per layer; prefetching
// further ahead or batching all loads changed nothing).  Here the head's q, k, v, dout (+ dout2), out slices (64-byte
// rows, pitch 20 floats), the graph's pe block and the statistics are staged ONCE with 16-byte requests; waves 0-3 take the
// query tiles (dq), waves 4-7 the key tiles (dk, dv), every operand - row layout or column gather - comes from LDS
// (conflict-free pitches), and the tile loops are plain runtime loops (nothing is held across tiles but the accumulators).
constexpr int kHbP = 16 + 4;   // pitch of a staged 16-float head row

__host__ __device__ inline int attn_bwd_head_lds_floats(int nt, bool pe_lds) {
  const int nr = 16 * nt;
  return 5 * nr * kHbP + (pe_lds ? nr * (nr + 4) : 0) + 3 * nr;
}

// PE_LDS: the graph's pe block staged in LDS (N <= 128: 66 KB); beyond that it no longer fits beside the operand tiles and
// is read from global memory where it is used - one 16-byte (or four 4-byte) request per tile pair in the dq role, four
// gathers in the dk / dv role, requested one tile ahead - everything else still comes from LDS (N <= 256).
template <int NT, bool PE_LDS>
__global__ __launch_bounds__(512) void attn_bwd_head_kernel(AttnArgs a) {
  constexpr int DH = 16, H = 4, P = kHbP, NR = 16 * NT, PEP = NR + 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lq = lane & 15, g = lane >> 4;
  const int w4 = wv & 3, role = wv >> 2;
  const int b = (int)blockIdx.x / H, h = (int)blockIdx.x % H, bh = blockIdx.x;
  const int n = a.n_real[b];
  const int KT = (n + 15) >> 4, NB = a.NB;
  float* Qs = feta_lds;        // [NR][P]
  float* Ks = Qs + NR * P;
  float* Vs = Ks + NR * P;
  float* Ds = Vs + NR * P;     // dout (+ dout2)
  float* Os = Ds + NR * P;     // out
  float* PE = Os + NR * P;     // [NR][PEP] (PE_LDS)
  float* ST = PE + (PE_LDS ? NR * PEP : 0);   // [NR][2]
  float* DL = ST + 2 * NR;     // [NR] delta
  const bool has_pe = a.pe != nullptr;
  const int nm1 = a.N - 1;
  const float* peg = has_pe ? a.pe + (int64_t)b * a.N * a.N : nullptr;   // this graph's block
  const bool pe_vec = (a.N & 3) == 0;

  constexpr int RI = (NR * 4 + 511) / 512;
  float4 qv[RI], kv[RI], vv[RI], dv_[RI], ov[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int idx = min(tid + 512 * i, NR * 4 - 1), node = min(idx >> 2, nm1), c4 = 4 * (idx & 3);
    qv[i] = *reinterpret_cast<const float4*>(tok_row(a.q, a.qsb, a.qsn, b, node, h, DH) + c4);
    kv[i] = *reinterpret_cast<const float4*>(tok_row(a.k, a.qsb, a.qsn, b, node, h, DH) + c4);
    vv[i] = *reinterpret_cast<const float4*>(tok_row(a.v, a.qsb, a.qsn, b, node, h, DH) + c4);
    ov[i] = *reinterpret_cast<const float4*>(tok_row(a.out, a.osb, a.osn, b, node, h, DH) + c4);
    float4 d1 = *reinterpret_cast<const float4*>(tok_row(a.dout, a.osb, a.osn, b, node, h, DH) + c4);
    if (a.dout2 != nullptr) {
      const float4 d2 = *reinterpret_cast<const float4*>(tok_row(a.dout2, a.osb, a.osn, b, node, h, DH) + c4);
      d1 = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
    }
    dv_[i] = d1;
  }
  float sv = 0.0f;
  if (tid < 2 * NR) sv = a.stats_in[((int64_t)bh * a.N + min(tid >> 1, nm1)) * 2 + (tid & 1)];
  // the graph's pe block: N x N contiguous elements, 16 requests in flight per thread
  const int nn = a.N * a.N;
  const float rn = 1.0f / (float)a.N;
  for (int base = tid; PE_LDS && base < nn; base += 16 * 512) {
    float pv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) pv[u] = has_pe ? a.pe[(int64_t)b * nn + min(base + 512 * u, nn - 1)] : 1.0f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = base + 512 * u;
      const int qq = (int)(((float)idx + 0.5f) * rn), kk = idx - qq * a.N;   // (idx + 1/2) / N: never near an integer
      if (idx < nn) PE[qq * PEP + kk] = pv[u];
    }
  }
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int idx = tid + 512 * i;
    if (idx < NR * 4) {
      const int off = (idx >> 2) * P + 4 * (idx & 3);   // rows >= N: copies of row N-1 (finite; masked below)
      *reinterpret_cast<float4*>(Qs + off) = qv[i];
      *reinterpret_cast<float4*>(Ks + off) = kv[i];
      *reinterpret_cast<float4*>(Vs + off) = vv[i];
      *reinterpret_cast<float4*>(Ds + off) = dv_[i];
      *reinterpret_cast<float4*>(Os + off) = ov[i];
    }
  }
  if (tid < 2 * NR) ST[tid] = sv;
  lds_barrier();
  if (tid < NR) {   // delta[q] = dout[q] . out[q]
    float d = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < DH; c4 += 4) {
      const float4 x = *reinterpret_cast<const float4*>(Ds + tid * P + c4), y = *reinterpret_cast<const float4*>(Os + tid * P + c4);
      d += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    if (tid < a.N) a.delta[(int64_t)bh * a.N + tid] = d;
    // per query, once: the reciprocal of the clamped normaliser (v_rcp_f32, 1 ulp) in place of the row sum, and delta
    // already zeroed where the clamp is active (the normaliser is a constant there) - the dk / dv waves used to divide
    // and select per (query, key) pair
    const float z = ST[2 * tid + 1];
    ST[2 * tid + 1] = fast_rcp(fmaxf(z, 1e-6f));
    DL[tid] = z < 1e-6f ? 0.0f : d;
  }
  lds_barrier();

  if (role == 0) {
    // dq of query tiles w4, w4 + 4: S^T orientation (key 4g+r, query lq)
    for (int qb = w4; qb < NB; qb += 4) {
      const int q = 16 * qb + lq;
      const bool qok = q < a.N;
      Feat<DH> qf, dof;
      load_row<DH>(qf, Qs + q * P, g, a.scale);
      load_row<DH>(dof, Ds + q * P, g);
      const float m = ST[2 * q], rinv = ST[2 * q + 1];
      const float delta = DL[q];
      f32x4 dq = zero4();
      float pnx[4] = {1.0f, 1.0f, 1.0f, 1.0f};
      const float* perow = has_pe ? peg + (int64_t)min(q, nm1) * a.N : nullptr;
      auto load_pe = [&](int kt_) {   // pe[q][16 kt_ + 4g .. + 3] (columns clamped: masked keys are selected away)
        if (!has_pe) return;
        const int c0 = 16 * kt_ + 4 * g;
        if (pe_vec && c0 + 3 <= nm1) {
          const float4 t = *reinterpret_cast<const float4*>(perow + c0);
          pnx[0] = t.x; pnx[1] = t.y; pnx[2] = t.z; pnx[3] = t.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) pnx[r] = perow[min(c0 + r, nm1)];
        }
      };
      if (!PE_LDS && KT > 0) load_pe(0);
      for (int kt = 0; kt < KT; ++kt) {
        Feat<DH> kf, vf;
        load_row<DH>(kf, Ks + (16 * kt + lq) * P, g);
        load_row<DH>(vf, Vs + (16 * kt + lq) * P, g);
        const f32x4 s = dot_rows<DH>(kf, qf, zero4());
        const f32x4 da = dot_rows<DH>(vf, dof, zero4());
        float pv[4];
        if (PE_LDS) {
          const float4 pe4 = *reinterpret_cast<const float4*>(PE + min(q, nm1) * PEP + 16 * kt + 4 * g);
          pv[0] = pe4.x; pv[1] = pe4.y; pv[2] = pe4.z; pv[3] = pe4.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = pnx[r];
          load_pe(min(kt + 1, KT - 1));   // the next tile's values travel under this tile's products
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * g + r;
          const float p = (key < n && qok) ? fast_exp(s[r] - m) * pv[r] * rinv : 0.0f;
          dq = mfma16(p * (da[r] - delta), Ks[key * P + lq], dq);   // (query 4g+r, c lq)
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = 16 * qb + 4 * g + r;
        if (qq < a.N) tok_row(a.dq, a.qsb, a.qsn, b, qq, h, DH)[lq] = dq[r] * a.scale;
      }
    }
  } else {
    // dk, dv of key tiles w4, w4 + 4: S orientation (query 4g+r, key lq)
    for (int kb = w4; kb < NB; kb += 4) {
      const int key = 16 * kb + lq;
      const bool kok = key < n;
      f32x4 dk = zero4(), dv = zero4();
      if (16 * kb < n) {
        Feat<DH> kf, vf;
        load_row<DH>(kf, Ks + key * P, g);
        load_row<DH>(vf, Vs + key * P, g);
        float pnx[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        const float* pecol = has_pe ? peg + min(key, nm1) : nullptr;
        auto load_pe = [&](int qb_) {   // pe[16 qb_ + 4g + r][key]
          if (!has_pe) return;
#pragma unroll
          for (int r = 0; r < 4; ++r) pnx[r] = pecol[(int64_t)min(16 * qb_ + 4 * g + r, nm1) * a.N];
        };
        if (!PE_LDS) load_pe(0);
        for (int qb = 0; qb < NB; ++qb) {
          Feat<DH> qf, dof;
          load_row<DH>(qf, Qs + (16 * qb + lq) * P, g, a.scale);
          load_row<DH>(dof, Ds + (16 * qb + lq) * P, g);
          const f32x4 s = dot_rows<DH>(qf, kf, zero4());
          const f32x4 da = dot_rows<DH>(dof, vf, zero4());
          float pv[4];
          if (!PE_LDS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = pnx[r];
            load_pe(min(qb + 1, NB - 1));
          }
          // (row max, reciprocal normaliser) and delta of this lane's four queries: three 16-byte LDS reads
          const float4 s01 = *reinterpret_cast<const float4*>(ST + 2 * (16 * qb + 4 * g));
          const float4 s23 = *reinterpret_cast<const float4*>(ST + 2 * (16 * qb + 4 * g) + 4);
          const float4 dl4 = *reinterpret_cast<const float4*>(DL + 16 * qb + 4 * g);
          const float mq[4] = {s01.x, s01.z, s23.x, s23.z}, rq[4] = {s01.y, s01.w, s23.y, s23.w};
          const float dq4[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qq = 16 * qb + 4 * g + r;
            const float m = mq[r], ri = rq[r];
            const bool ok = qq < a.N && kok;
            if (PE_LDS) pv[r] = PE[min(qq, nm1) * PEP + min(key, nm1)];
            const float p = ok ? fast_exp(s[r] - m) * pv[r] * ri : 0.0f;
            const float ds = p * (da[r] - dq4[r]);
            dv = mfma16(p, Ds[qq * P + lq], dv);              // (key 4g+r, c lq)
            dk = mfma16(ds, Qs[qq * P + lq] * a.scale, dk);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = 16 * kb + 4 * g + r;
        if (kk < a.N) {
          tok_row(a.dk, a.qsb, a.qsn, b, kk, h, DH)[lq] = dk[r];
          tok_row(a.dv, a.qsb, a.qsn, b, kk, h, DH)[lq] = dv[r];
        }
      }
    }
  }
}

template <int NT, bool PE_LDS>
void launch_bwd_head_t(const AttnArgs& a, hipStream_t stream) {
  const size_t lds = sizeof(float) * attn_bwd_head_lds_floats(NT, PE_LDS);
  auto kern = attn_bwd_head_kernel<NT, PE_LDS>;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  hipLaunchKernelGGL(kern, dim3(a.B * a.H), dim3(512), lds, stream, a);
}

// -> true if the one-workgroup-per-(graph, head) backward was launched (4 heads x dh 16, 64 < N <= 256)
bool try_bwd_head(const AttnArgs& a, int dh, hipStream_t stream) {
  if (a.H != 4 || dh != 16 || a.N <= 64 || a.N > 256) return false;
  if (const char* e = getenv("FETA_ATTN_BWD_HEAD"))
    if (e[0] == '0') return false;
  switch (a.NB) {
    case 5: launch_bwd_head_t<5, true>(a, stream); break;
    case 6: launch_bwd_head_t<6, true>(a, stream); break;
    case 7: launch_bwd_head_t<7, true>(a, stream); break;
    case 8: launch_bwd_head_t<8, true>(a, stream); break;
    default:
      if (a.NB <= 12) launch_bwd_head_t<12, false>(a, stream);
      else launch_bwd_head_t<16, false>(a, stream);
  }
  return true;
}

// -> true if the one-workgroup-per-graph backward was launched (4 heads x dh 16, N <= 64)
bool try_bwd_graph(const AttnArgs& a, int dh, hipStream_t stream) {
  if (a.H != 4 || dh != 16 || a.N > 64) return false;
  // measured (MI355X, N = 37): per-(head, tile) waves win while a batch cannot fill the chip (12.6 vs 13.1 us
  // at 128 graphs), one workgroup per graph wins from 256 graphs up (13.7 vs 17.4 us; 460 vs 880 us at
  // 16384).  FETA_ATTN_BWD_GRAPH=0 / 1 forces the choice (tests).
  const char* e = getenv("FETA_ATTN_BWD_GRAPH");
  if (e != nullptr ? e[0] == '0' : a.B < 192) return false;
  switch (a.NB) {
    case 1: launch_bwd_graph_t<1>(a, stream); break;
    case 2: launch_bwd_graph_t<2>(a, stream); break;
    case 3: launch_bwd_graph_t<3>(a, stream); break;
    default: launch_bwd_graph_t<4>(a, stream); break;
  }
  return true;
}

template <int DH>
int launch_bwd(const AttnArgs& a, hipStream_t stream) {
  if (try_bwd_graph(a, DH, stream)) return check_launch("feta_attn_bwd");
  if (try_bwd_head(a, DH, stream)) return check_launch("feta_attn_bwd");
  if constexpr (DH <= 16) {
    if (a.NB <= 3) { launch_bwd_dense_t<DH, 3>(a, stream); return check_launch("feta_attn_bwd"); }
    if (a.NB <= 4) { launch_bwd_dense_t<DH, 4>(a, stream); return check_launch("feta_attn_bwd"); }
    // (measured, PATTERN B = 64, N = 128: the batched-load form with 8 key tiles - 256 + 78 registers, one wave per SIMD -
    // 0.514 ms per step against 0.481 for the two pipelined kernels below; prefetching two tiles ahead in those: 0.489.
    // Their 10 / 26 gather instructions per tile are bound by the address unit, not by one memory latency)
  }
  FETA_REQUIRE(a.dout2 == nullptr, "attn_bwd: dout2 is only supported for N <= 64, dh <= 16");
  const dim3 grid((a.total + kWaves - 1) / kWaves), block(64 * kWaves);
  auto k1 = attn_bwd_dq_kernel<DH>;
  hipLaunchKernelGGL(k1, grid, block, 0, stream, a);
  auto k2 = attn_bwd_dkdv_kernel<DH>;
  hipLaunchKernelGGL(k2, grid, block, 0, stream, a);
  return check_launch("feta_attn_bwd");
}

int check_common(const void* q, const void* k, const void* v, int64_t sb, int64_t sn, int64_t osb,
                 int64_t osn, int B, int N, int H, int dh) {
  FETA_REQUIRE(B > 0 && N > 0 && H > 0, "attn: empty shape B=%d N=%d H=%d", B, N, H);
  FETA_REQUIRE(N <= FETA_MAX_NODES, "attn: N=%d exceeds FETA_MAX_NODES", N);
  FETA_REQUIRE(dh == 4 || dh == 8 || dh == 16 || dh == 32 || dh == 64,
               "attn: head dim %d not in {4,8,16,32,64}", dh);
  FETA_REQUIRE((sb % 4) == 0 && (sn % 4) == 0 && (osb % 4) == 0 && (osn % 4) == 0,
               "attn: strides must be multiples of 4 elements");
  FETA_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v), "attn: q/k/v must be 16-byte aligned");
  return FETA_OK;
}

}  // namespace feta

using namespace feta;

#define FETA_DISPATCH_DH(dh, CALL) \
  switch (dh) {                    \
    case 4: return CALL(4);        \
    case 8: return CALL(8);        \
    case 16: return CALL(16);      \
    case 32: return CALL(32);      \
    default: return CALL(64);      \
  }

extern "C" int feta_attn_fwd(const float* q, const float* k, const float* v, int64_t qkv_sb,
                             int64_t qkv_sn, const float* pe, const int32_t* n_real, float* out,
                             int64_t o_sb, int64_t o_sn, float* attn, float* stats, float scale,
                             int B, int N, int H, int dh, feta_stream_t stream) {
  int rc = check_common(q, k, v, qkv_sb, qkv_sn, o_sb, o_sn, B, N, H, dh);
  if (rc != FETA_OK) return rc;
  FETA_REQUIRE(out != nullptr && stats != nullptr && n_real != nullptr, "attn_fwd: null output");
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.pe = pe; a.n_real = n_real;
  a.out_w = out; a.attn = attn; a.stats = stats;
  a.qsb = qkv_sb; a.qsn = qkv_sn; a.osb = o_sb; a.osn = o_sn;
  a.scale = scale; a.B = B; a.N = N; a.H = H; a.NB = (N + 15) / 16;
  a.total = B * H * a.NB;
  const int kt_max = a.NB;
#define CALL(D) launch_fwd<D>(a, kt_max, (hipStream_t)stream)
  FETA_DISPATCH_DH(dh, CALL)
#undef CALL
}

extern "C" int feta_attn_bwd(const float* q, const float* k, const float* v, int64_t qkv_sb,
                             int64_t qkv_sn, const float* pe, const int32_t* n_real,
                             const float* out, const float* dout, const float* dout2, int64_t o_sb,
                             int64_t o_sn, const float* stats, float* delta, float* dq, float* dk,
                             float* dv, float scale, int B, int N, int H, int dh, feta_stream_t stream) {
  int rc = check_common(q, k, v, qkv_sb, qkv_sn, o_sb, o_sn, B, N, H, dh);
  if (rc != FETA_OK) return rc;
  FETA_REQUIRE(out && dout && stats && delta && dq && dk && dv && n_real, "attn_bwd: null pointer");
  FETA_REQUIRE(aligned16(out) && aligned16(dout) && aligned16(dq) && aligned16(dk) && aligned16(dv) &&
                   (!dout2 || aligned16(dout2)),
               "attn_bwd: pointers must be 16-byte aligned");
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.pe = pe; a.n_real = n_real;
  a.out = out; a.dout = dout; a.dout2 = dout2; a.stats_in = stats; a.delta = delta;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.qsb = qkv_sb; a.qsn = qkv_sn; a.osb = o_sb; a.osn = o_sn;
  a.scale = scale; a.B = B; a.N = N; a.H = H; a.NB = (N + 15) / 16;
  a.total = B * H * a.NB;
#define CALL(D) launch_bwd<D>(a, (hipStream_t)stream)
  FETA_DISPATCH_DH(dh, CALL)
#undef CALL
}
