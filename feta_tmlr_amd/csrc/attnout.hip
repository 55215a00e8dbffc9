// A1 for graphs beyond the one-launch attention block (64 < N <= 256; d = 64 = 4 heads x 16): attention core +
// out_proj + degree scale + residual + BatchNorm statistics of DiffTransformerEncoderLayer as ONE launch behind the
// in_proj launch (contract transformer/models.py:166-167,179,244; config 4 - PATTERN, N_pad ~ 120-190 - entry
// experiments/run_transformer_gengcn_SBM_cv.py, node-level head transformer/models.py:1069-1071).  Replaces
// feta_attn_fwd -> feta_rowlin_fwd_ex (out_proj) of the op-by-op sequence: the per-head outputs of a 32-row query chunk
// meet in an LDS tile that is the out_proj operand, so `out` is written once and never read back, and the general
// attention kernel's 4-byte gathers of pe and V (one wave per (graph, head, 16 queries), everything from global
// memory) become one coalesced stream per workgroup.
//
// Work decomposition: one workgroup per (graph b, chunk of 32 query rows), 8 waves = (head h, query tile p of the
// chunk).  A workgroup stages, once, the V rows of the WHOLE graph (all heads: [N][64], the P.V operand is gathered
// from it conflict-free) and the chunk's pe rows; K^T operands come straight from global memory in their operand
// layout (16 bytes per lane and key tile); W_out rows likewise (a wave needs the 16 rows of its output columns only -
// nothing is shared between waves, so nothing is staged).  The chunks of a graph are workgroups b, b + B, ...: the same
// XCD (the same L2) when B is a multiple of 8.
#include <cstdlib>

#include "feta_abi_common.h"
#include "feta_colsum.h"
#include "feta_lp.h"
#include "feta_rowops.h"

namespace feta {

typedef feta_attn_block AoArgs;   // include/feta_hip.h (w_in, b_in, x_stats unused: the in_proj launch came first)

constexpr int kAoThreads = 512, kAoD = 64, kAoH = 4, kAoDH = 16, kAoChunk = 32;
constexpr int kAoStgKeys = 64;    // keys per pass of the staged attn write

template <class T>
__host__ __device__ inline int attn_out_lds_bytes(int ktm, bool attn) {
  const int nr = 16 * ktm, P = kAoD + Lp<T>::PAD;
  int b = (int)sizeof(T) * (nr * P + kAoChunk * P);                    // V rows of the graph, OUT tile of the chunk
  int f = 128 + kAoChunk * (nr + 4);                                    // statistics hand-over, pe rows of the chunk
  if (attn) f += 8 * 16 * (kAoStgKeys + 1);                             // per-wave probability staging
  return b + 4 * f;
}

// fp32 master weights -> the row operand of a 64-wide contraction (rounded here for bf16 tiles)
template <class T>
__device__ __forceinline__ void load_w_row_op(RowOp<T, kAoD>& t, const float* row, int g) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 x = *reinterpret_cast<const float4*>(row + 16 * j + 4 * g);
    t.o[j] = Lp<T>::mk(x.x, x.y, x.z, x.w);
  }
}

template <class T, int KTM>
__global__ __launch_bounds__(kAoThreads) void attn_out_fwd_kernel(AoArgs a, ColsumPlan sums, int main_grid) {
  // workgroups beyond main_grid reduce column sums (feta_colsum.h): s = colsum(gcn.weight) of the coefficient generator - a
  // function of the parameters alone - rides in the first launch of a forward pass instead of a launch of its own
  if ((int)blockIdx.x >= main_grid) {
    colsum_role<kAoThreads>(sums, (int)blockIdx.x - main_grid);
    return;
  }
  typedef Lp<T> L;
  typedef typename L::Op Op;
  typedef typename L::Vec Vec;
  constexpr int TH = kAoThreads, D = kAoD, DH = kAoDH, P = D + L::PAD, NR = 16 * KTM, CH = kAoChunk;
  constexpr int RV = D / L::VEC;                       // 16-byte vectors of a 64-element row
  constexpr int VI = (NR * RV + TH - 1) / TH;          // ... of the graph's V rows, per thread
  constexpr int PEP = NR + 4, PEI = (CH * NR + TH - 1) / TH;
  constexpr int SK = kAoStgKeys, SP = SK + 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = wv & 3, p = wv >> 2, lq = lane & 15, g = lane >> 4;
  const int b = (int)blockIdx.x % a.B, ch = (int)blockIdx.x / a.B;
  const int q0 = CH * ch;                              // first query row of the chunk
  const int qt = q0 + 16 * p;                          // first row of this wave's query tile
  const bool qvalid = qt < a.N;                        // (wave-uniform; the chunk's second tile may lie beyond N)
  T* Vs = reinterpret_cast<T*>(lds_bytes());           // [NR][P]
  T* Os = Vs + NR * P;                                 // [CH][P]
  float* sx = reinterpret_cast<float*>(Os + CH * P);   // [4 heads][4 g][8]
  float* Pe = sx + 128;                                // [CH][PEP]
  float* stg = Pe + CH * PEP + wv * 16 * SP;           // [16][SP], only when attn is written
  const T* gqkv = reinterpret_cast<const T*>(a.qkv);
  const T* gpe = reinterpret_cast<const T*>(a.pe);
  const T* gx = reinterpret_cast<const T*>(a.x);
  T* gout = reinterpret_cast<T*>(a.out);
  T* gy = reinterpret_cast<T*>(a.y);
  const int n = a.n_real[b];
  const int KT = (n + 15) >> 4, nm1 = a.N - 1;
  const int koff = a.tie_qk ? 0 : D;
  const int64_t rb = (int64_t)b * a.row_sb, rsn = a.row_sn;
  auto rowof = [rb, rsn](int node) { return rb + (int64_t)node * rsn; };   // (captures copies: csrc lesson on [&] and kernel arguments)

  // ---- every request of the workgroup goes out before the first one is consumed -----------------------------------
  Vec vv[VI];
#pragma unroll
  for (int i = 0; i < VI; ++i) {
    const int idx = min(tid + TH * i, NR * RV - 1), node = idx / RV, q = idx % RV;
    vv[i] = L::ldv(gqkv + rowof(min(node, nm1)) * 3 * D + 2 * D + L::VEC * q);
  }
  const bool has_pe = a.pe != nullptr;
  const int q1 = min(a.N, q0 + CH), pecnt = (q1 - q0) * a.N;
  const int64_t pebase = (int64_t)b * a.N * a.N + (int64_t)q0 * a.N;
  float pel[PEI];
#pragma unroll
  for (int i = 0; i < PEI; ++i) pel[i] = has_pe ? L::ld1(gpe + pebase + min(tid + TH * i, pecnt - 1)) : 1.0f;
  const int qn = min(qt + lq, nm1);                    // this lane's query row (clamped)
  const Op qs = L::ld_scaled(gqkv + rowof(qn) * 3 * D + DH * h + 4 * g, a.scale);
  Op kf[KTM];
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt)
    kf[kt] = L::ld(gqkv + rowof(min(16 * kt + lq, nm1)) * 3 * D + koff + DH * h + 4 * g);
  RowOp<T, D> wf;
  load_w_row_op<T>(wf, a.w_out + (DH * h + lq) * D, g);
  const int o0 = DH * h + 4 * g;
  float4 bo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), ks = bo, xsc = make_float4(1.0f, 1.0f, 1.0f, 1.0f), xsh = bo;
  if (a.b_out != nullptr) bo = *reinterpret_cast<const float4*>(a.b_out + o0);
  if (a.y_shift != nullptr) ks = *reinterpret_cast<const float4*>(a.y_shift + o0);
  if (a.x_bn != nullptr) {
    xsc = *reinterpret_cast<const float4*>(a.x_bn + o0);
    xsh = *reinterpret_cast<const float4*>(a.x_bn + D + o0);
  }
  const float rs = a.rowscale != nullptr ? a.rowscale[rowof(qn)] : 1.0f;
  float res[4];
  L::ld4(gx + rowof(qn) * D + o0, res);

  // ---- V rows and the chunk's pe rows -> LDS ----------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < VI; ++i) {
    const int idx = tid + TH * i, node = idx / RV, q = idx % RV;
    if (VI * TH != NR * RV && idx >= NR * RV) continue;
    L::stv(Vs + node * P + L::VEC * q, vv[i]);        // rows >= N: a copy of row N-1 (finite; their probabilities are 0)
  }
  {
    const float rn = 1.0f / (float)a.N;
#pragma unroll
    for (int i = 0; i < PEI; ++i) {
      const int idx = tid + TH * i;
      const int qq = (int)(((float)idx + 0.5f) * rn), kk = idx - qq * a.N;   // (idx + 1/2) / N: never near an integer
      if (idx < pecnt) Pe[qq * PEP + kk] = pel[i];
    }
  }
  lds_barrier();

  // ---- attention core of this wave's query tile (transposed score tile: rows = keys) ------------------------------
  const int bh = b * kAoH + h;
  // (scalars, not f32x4 acc[KTM]: the optimizer promotes such an array to ONE 4 KTM-wide vector value, and every
  // conditional tile update then copies the whole tuple - 260 bytes of scratch per lane and 37 us at KTM = 8)
  float acc[KTM][4];
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt) {
    f32x4 t = zero4();
    if (kt < KT && qvalid) t = L::mma(kf[kt], qs, zero4());   // (key 4g+r, query lq)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[kt][r] = t[r];
  }
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * kt + 4 * g + r < n) m = fmaxf(m, acc[kt][r]);
  m = fmaxf(m, shfl_xor(m, 16));
  m = fmaxf(m, shfl_xor(m, 32));
  float z = 0.0f;
  const float* perow = Pe + min(16 * p + lq, max(q1 - q0 - 1, 0)) * PEP;
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt) {
    const float4 t = *reinterpret_cast<const float4*>(perow + 16 * kt + 4 * g);
    const float pv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool kok = 16 * kt + 4 * g + r < n && qvalid;
      const float e = kok ? fast_exp(acc[kt][r] - m) * pv[r] : 0.0f;
      acc[kt][r] = e;
      z += e;
    }
  }
  z += shfl_xor(z, 16);
  z += shfl_xor(z, 32);
  const float rinv = 1.0f / fmaxf(z, 1e-6f);
  if (g == 0 && qvalid && qt + lq < a.N) {
    float* st = a.attn_stats + ((int64_t)bh * a.N + qt + lq) * 2;
    st[0] = m;
    st[1] = z;
  }
  f32x4 o = zero4();
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[kt][r] *= rinv;
    if (kt >= KT || !qvalid) continue;
    const Op vb = L::gather(Vs + (16 * kt + 4 * g) * P + DH * h + lq, P);   // (key 4g+s, c' lq)
    o = L::mma(L::mk(acc[kt][0], acc[kt][1], acc[kt][2], acc[kt][3]), vb, o);   // (query 4g+r, c' lq)
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) L::st1(Os + (16 * p + 4 * g + r) * P + DH * h + lq, o[r]);
  if (a.attn != nullptr && qvalid) {
    // attn[b, h, qt .. qt+15, :] in passes of 64 keys: stage, then whole 256-byte row segments
    const int rows = min(16, a.N - qt);
    float* dst = a.attn + ((int64_t)bh * a.N + qt) * a.N;
#pragma unroll
    for (int ps = 0; ps < (KTM + 3) / 4; ++ps) {
      if (SK * ps >= a.N) break;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kt = 4 * ps + j;
        if (kt >= KTM) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[lq * SP + 16 * j + 4 * g + r] = acc[kt][r];
      }
      wave_lds_sync();
      const int cols = min(SK, a.N - SK * ps);
#pragma unroll 4
      for (int qq = 0; qq < 16; ++qq)
        if (qq < rows && lane < cols) dst[(int64_t)qq * a.N + SK * ps + lane] = stg[qq * SP + lane];
      wave_lds_sync();
    }
  }
  lds_barrier();

  // ---- concat rows of the chunk to HBM (whole rows); out_proj + degree + residual + statistics --------------------
  {
    const int node = q0 + tid / RV, q = tid % RV;
    if (tid < CH * RV && node < a.N) {
      const Vec ov = L::ldv(Os + (tid / RV) * P + L::VEC * q);
      L::stv(gout + rowof(node) * D + L::VEC * q, ov);
      if (a.out_f32 != nullptr) {
        float f[L::VEC];
        L::unpack(ov, f);
#pragma unroll
        for (int e = 0; e < L::VEC; e += 4)
          *reinterpret_cast<float4*>(a.out_f32 + rowof(node) * D + L::VEC * q + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
      }
    }
  }
  float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (qvalid) {
    const bool rok = qt + lq < a.N;
    RowOp<T, D> of;
    load_row_op<T, D>(of, Os + (16 * p + lq) * P, g);
    const f32x4 t = dot_row_ops<T, D>(wf, of, zero4());   // (o = 16h + 4g + r, node lq)
    const float rx[4] = {res[0] * xsc.x + xsh.x, res[1] * xsc.y + xsh.y, res[2] * xsc.z + xsh.z, res[3] * xsc.w + xsh.w};
    const float v[4] = {(t[0] + bo.x) * rs + rx[0], (t[1] + bo.y) * rs + rx[1], (t[2] + bo.z) * rs + rx[2],
                        (t[3] + bo.w) * rs + rx[3]};
    if (rok) L::st4(gy + rowof(qn) * D + o0, v[0], v[1], v[2], v[3]);
    const float kv[4] = {ks.x, ks.y, ks.z, ks.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x1 = rok ? v[r] - kv[r] : 0.0f;
      s1[r] = row16_sum(x1);
      s2[r] = row16_sum(x1 * x1);
    }
  }
  if (a.y_stats != nullptr) {
    // the two tiles of the chunk hold sums over different rows of the same columns: the second hands over, the first adds
    if (p == 1 && lq == 0) {
      float* e = sx + (4 * h + g) * 8;
      *reinterpret_cast<float4*>(e) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(e + 4) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    }
    lds_barrier();
    if (p == 0 && lq == 0) {
      const float* e = sx + (4 * h + g) * 8;
      const float4 t1 = *reinterpret_cast<const float4*>(e), t2 = *reinterpret_cast<const float4*>(e + 4);
      float* st = a.y_stats + (int64_t)blockIdx.x * 2 * D;
      *reinterpret_cast<float4*>(st + o0) = make_float4(s1[0] + t1.x, s1[1] + t1.y, s1[2] + t1.z, s1[3] + t1.w);
      *reinterpret_cast<float4*>(st + D + o0) = make_float4(s2[0] + t2.x, s2[1] + t2.y, s2[2] + t2.z, s2[3] + t2.w);
      if (blockIdx.x == 0) *reinterpret_cast<float4*>(a.y_stats + (int64_t)main_grid * 2 * D + o0) = ks;   // the shift row
    }
  }
}

template <class T, int KTM>
int launch_attn_out(const AoArgs& a, const feta_colsum_seg* segs, int nseg, hipStream_t stream) {
  size_t lds = attn_out_lds_bytes<T>(KTM, a.attn != nullptr);
  ColsumPlan plan{};
  const int tiles = plan_colsum(segs, nseg, plan);
  if (tiles > 0 && lds < sizeof(float) * colsum_role_lds_floats(kAoThreads)) lds = sizeof(float) * colsum_role_lds_floats(kAoThreads);
  auto kern = attn_out_fwd_kernel<T, KTM>;
  static LdsSeen lds_seen;
  allow_dynamic_lds(kern, lds, lds_seen);
  const int chunks = (a.N + kAoChunk - 1) / kAoChunk;
  hipLaunchKernelGGL(kern, dim3(a.B * chunks + tiles), dim3(kAoThreads), lds, stream, a, plan, a.B * chunks);
  return check_launch("feta_attn_out_fwd");
}

template <class T>
int dispatch_attn_out(const AoArgs& a, const feta_colsum_seg* segs, int nseg, hipStream_t stream) {
  const int kt = (a.N + 15) / 16;
  if (kt <= 4) return launch_attn_out<T, 4>(a, segs, nseg, stream);
  if (kt <= 8) return launch_attn_out<T, 8>(a, segs, nseg, stream);
  if (kt <= 12) return launch_attn_out<T, 12>(a, segs, nseg, stream);
  return launch_attn_out<T, 16>(a, segs, nseg, stream);
}

}  // namespace feta

using namespace feta;

extern "C" int feta_attn_out_supported(int N, int d_model, int heads) {
  return (d_model == kAoD && heads == kAoH && N >= 1 && N <= 256) ? 1 : 0;
}

extern "C" int feta_attn_out_stat_rows(int B, int N) {
  if (B < 1 || N < 1 || N > 256) return 0;
  return B * ((N + kAoChunk - 1) / kAoChunk);
}

extern "C" int feta_attn_out_fwd(const feta_attn_block* d, feta_stream_t stream) {
  return feta_attn_out_fwd_sums(d, nullptr, 0, stream);
}

extern "C" int feta_attn_out_fwd_sums(const feta_attn_block* d, const feta_colsum_seg* segs, int nseg, feta_stream_t stream) {
  FETA_REQUIRE(d != nullptr, "attn_out_fwd: null descriptor");
  FETA_REQUIRE(nseg >= 0 && nseg <= FETA_COLSUM_MAX_SEGS && (nseg == 0 || segs != nullptr),
               "attn_out_fwd: 0..%d column-sum segments", FETA_COLSUM_MAX_SEGS);
  for (int i = 0; i < nseg; ++i) FETA_REQUIRE(colsum_seg_ok(segs[i]), "attn_out_fwd: bad segment %d", i);
  const AoArgs& a = *d;
  FETA_REQUIRE(a.x && a.w_out && a.n_real && a.qkv && a.out && a.attn_stats && a.y, "attn_out_fwd: null pointer");
  FETA_REQUIRE(a.B > 0 && a.N >= 1 && a.N <= 256, "attn_out_fwd: N=%d outside [1,256]", a.N);
  FETA_REQUIRE(a.M == a.B * a.N, "attn_out_fwd: M=%d is not B*N", a.M);
  FETA_REQUIRE(a.x_stats == nullptr, "attn_out_fwd: x is seen through a published parameter block (x_bn) only");
  FETA_REQUIRE(aligned16(a.x) && aligned16(a.w_out) && aligned16(a.qkv) && aligned16(a.out) && aligned16(a.y) &&
               aligned16(a.y_stats) && aligned16(a.b_out) && aligned16(a.out_f32) && aligned16(a.y_shift) &&
               aligned16(a.x_bn),
               "attn_out_fwd: tensors must be 16-byte aligned");
  // (the kernel is written against the storage policy of feta_lp.h, but the bf16 layer stack stops at N <= 64 - shapes
  // beyond run its op-by-op path - so only the fp32 instantiation is built and tested)
  FETA_REQUIRE(a.dtype == FETA_F32, "attn_out_fwd: fp32 token tensors only (dtype %d)", a.dtype);
  return dispatch_attn_out<float>(a, segs, nseg, (hipStream_t)stream);
}
