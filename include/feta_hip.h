/* feta_hip.h - C ABI of libfeta_hip.so: the MI355X (gfx950) kernels behind the FeTA
 * spectral-attention block (dense per-graph attention with a multiplicative
 * positional kernel -> attention-conditioned filter coefficients -> dynamic
 * Chebyshev / eigenbasis spectral filter), forward and backward.
 *
 * The reference (ansonb/FeTA_TMLR) has no FFI for this path - it is ~40 PyTorch /
 * torch-geometric ops per layer - so each entry point below cites the reference
 * Python text whose arithmetic it replaces (paths relative to the reference root).
 * The Python side (feta_tmlr_amd/_abi.py) binds exactly these symbols with ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller and valid until the
 *    given stream reaches the enqueue point; nothing is allocated or synchronised
 *    inside; kernels are stream-ordered and graph-capturable;
 *  - floating-point data is fp32 unless an entry point or a descriptor's `dtype` says bf16, row-major; `stream` is a
 *    hipStream_t;
 *  - "token tensors" (q/k/v, per-head outputs, filter input/output) are addressed
 *    as ptr[b*sb + i*sn + h*dh + c] with element strides sb (graph) and sn (node),
 *    so seq-first [N,B,d] and batch-first [B,N,d] storage both work without copies;
 *    base pointers and strides must be multiples of 4 elements (16 bytes);
 *  - n_real[b] (int32) is the number of real nodes of graph b, 1 <= n_real[b] <= N;
 *    padding is the suffix i >= n_real[b] (reference collate: transformer/data.py:210);
 *  - return 0 on success, a negative FETA_E_* code otherwise; feta_last_error()
 *    gives the message for the calling thread.
 */
#ifndef FETA_HIP_H_
#define FETA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FETA_ABI_VERSION 11

#define FETA_OK 0
#define FETA_E_ARG (-1)     /* bad shape / stride / alignment / unsupported size */
#define FETA_E_LAUNCH (-2)  /* the HIP runtime rejected a launch */

typedef void* feta_stream_t;

/* Storage type of the token tensors of the fused layer-stack descriptors (feta_attn_block, feta_attn_block_grad,
 * feta_ffn, feta_ffn_grad; ABI 7).  FETA_BF16: the tensors marked [T] in each descriptor hold bf16 (the pointers stay
 * typed float* in the structs; rows must be 16-byte aligned), the kernels stage bf16 tiles and run their contractions
 * on v_mfma_f32_16x16x16_bf16; weights, biases, BatchNorm parameter blocks and partial sums, softmax statistics, the
 * attention matrix, and every weight / bias gradient stay fp32.  The reference has no reduced-precision mode
 * (experiments/run_transformer_gengcn.py:115-164): this is the storage path of BASELINE configs 3 and 5. */
#define FETA_F32 0
#define FETA_BF16 1

int feta_version(void);
const char* feta_last_error(void);

/* Largest padded node count and head dim the tile templates are instantiated for. */
#define FETA_MAX_NODES 256
#define FETA_MAX_HEAD_DIM 64

/* ---- A1: attention core ---------------------------------------------------------
 * Replaces the score/softmax/weighted-sum of DiffTransformerEncoderLayer.self_attn
 * (source absent from the reference; contract from transformer/models.py:166-167,179,
 * 244,275; form witnesses LSPE/layers/graphit_gt_layer.py:39-43,120-131,164):
 *   s = scale * q.k^T ; keys >= n_real masked ; e = exp(s - rowmax) ; e *= pe (if pe) ;
 *   attn = e / max(rowsum(e), 1e-6) ; out = attn . v
 * q,k,v   token tensors (strides qkv_sb, qkv_sn), all N query rows are computed
 * pe      [B,N,N] or NULL (broadcast over heads)
 * out     token tensor (strides o_sb, o_sn): out_each_head[b,i,h,:]
 * attn    [B,H,N,N] or NULL (skip the write; backward recomputes from stats)
 * stats   [B,H,N,2]: (rowmax, un-clamped rowsum) per query row
 */
int feta_attn_fwd(const float* q, const float* k, const float* v,
                  int64_t qkv_sb, int64_t qkv_sn,
                  const float* pe, const int32_t* n_real,
                  float* out, int64_t o_sb, int64_t o_sn,
                  float* attn, float* stats, float scale,
                  int B, int N, int H, int dh, feta_stream_t stream);

/* Backward of feta_attn_fwd w.r.t. q,k,v given dout (same strides as out).
 * delta [B,H,N] is caller-provided scratch (rowsum(dout*out)).
 * dq,dk,dv use the q/k/v strides and are fully written (zeros on padded keys).
 * dout2 (nullable; N <= 64 and dh <= 16 only) is a second gradient into the same output, added to
 * dout on load (the filter path's gradient into out_each_head).
 * The gradient through the rowmax subtraction is dropped (it is exactly zero
 * unless the 1e-6 clamp is active). */
int feta_attn_bwd(const float* q, const float* k, const float* v,
                  int64_t qkv_sb, int64_t qkv_sn,
                  const float* pe, const int32_t* n_real,
                  const float* out, const float* dout, const float* dout2, int64_t o_sb, int64_t o_sn,
                  const float* stats, float* delta,
                  float* dq, float* dk, float* dv, float scale,
                  int B, int N, int H, int dh, feta_stream_t stream);

/* ---- A2: filter-coefficient generator ---------------------------------------------
 * Replaces DiffTransformerEncoderGenGCN.get_filter_coefficients up to (not including)
 * self.linear: transformer/models.py:240-283 with GCNConv semantics of
 * transformer/GenGCN.py:55-102,393-402, using the exact collapse
 * GCNConv(ones)[j] = c_j * colsum(W) + b   (SURVEY F7).
 *   block (h,b): w = attn[b,h,:n,:n] ; w_jj = 1 where attn_jj == 0 ;
 *   deg_j = sum_i w_ij ; c_j = sum_i deg_i^-1/2 w_ij deg_j^-1/2 ;
 *   pooled[h*B+b, :] = mean_j tanh(c_j * s + gcn_bias)
 * cj      [H*B, N] out (saved for backward; rows >= n are zero)
 * pooled  [H*B, C] out
 */
int feta_coeff_fwd(const float* attn, const int32_t* n_real,
                   const float* s, const float* gcn_bias,
                   float* cj, float* pooled,
                   int B, int N, int H, int C, feta_stream_t stream);

/* ds[c] = sum_{blk,j} dpooled*(1-z^2)*c_j/n ; dbias[c] likewise without c_j.
 * partial [G, 2, C] scratch with G = feta_coeff_bwd_groups(B,H); when dbias == ds + C both are
 * reduced by one launch.  ds == NULL (ABI 6): no reduction launch - the caller reduces the partials ([G, 2C]: ds from
 * columns 0..C-1, dbias from C..2C-1) with feta_colsum_multi, together with whatever else it has pending.  dw_dense (nullable) [dw_rows, C]: the gradient of gcn.weight itself - s = 1^T W,
 * so every row of dW equals ds; the reduction launch writes the rows (no broadcast copy afterwards). */
int feta_coeff_bwd_groups(int B, int H);
int feta_coeff_bwd(const float* cj, const int32_t* n_real,
                   const float* s, const float* gcn_bias, const float* dpooled,
                   float* partial, float* ds, float* dbias, float* dw_dense, int dw_rows,
                   int B, int N, int H, int C, feta_stream_t stream);

/* out[c] = sum_r in[r, c]  (s = colsum(gcn.weight); also reduces per-block partials) */
int feta_colsum(const float* in, float* out, int R, int C, feta_stream_t stream);

/* Several independent column sums in one launch (bias gradients that become available together):
 * out[c] = sum_r in[r*ld + c], r < R, c < C (ld = 0: C); bcast_out (nullable) [bcast_rows, C] additionally
 * receives the result in every row. */
#define FETA_COLSUM_MAX_SEGS 12
typedef struct feta_colsum_seg {
  const float* in;
  float* out;
  int R, C, ld;
  float* bcast_out;
  int bcast_rows;
} feta_colsum_seg;
int feta_colsum_multi(const feta_colsum_seg* segs, int nseg, feta_stream_t stream);

/* ---- the C x C linear of the coefficient generator (ABI 6) ----
 * Replaces self.linear of DiffTransformerEncoderGenGCN (transformer/models.py:284; nn.Linear(C, C) applied to the
 * pooled [H*B, C] rows) and its autograd backward: y = x w^T + bias as one launch, and dx = dy w, dw = dy^T x,
 * db = colsum(dy) as ONE launch (workgroups take roles) that can also carry up to FETA_COLSUM_MAX_SEGS pending column
 * sums of the caller in trailing workgroups.  x [R,K], w [N,K], y / dy [R,N]; exact-fp32 MFMA, deterministic.
 * feta_lin_supported: R, K, N multiples of 4 and K, N >= 16 (else the caller uses a library GEMM).
 * dx and db may be NULL. */
int feta_lin_supported(int R, int K, int N);
int feta_lin_fwd(const float* x, const float* w, const float* bias, float* y, int R, int K, int N,
                 feta_stream_t stream);
int feta_lin_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int R, int K,
                 int N, const feta_colsum_seg* segs, int nseg, feta_stream_t stream);
/* the same with a compute type (ABI 7): FETA_BF16 = the fp32 operands are rounded to bf16 when they are staged and the
 * products run on v_mfma_f32_16x16x16_bf16 with fp32 accumulation (bf16 storage legs); outputs, bias gradient and the
 * pending column sums stay fp32.  Shapes whose dims are multiples of 64 (256 for K, N) take the LDS-tiled kernels -
 * one GFLOP per product at the BASELINE shape (R = 512, K = N = C = 1024), where they replace the library GEMMs. */
int feta_lin_fwd_ex(const float* x, const float* w, const float* bias, float* y, int R, int K, int N, int compute,
                    feta_stream_t stream);
int feta_lin_bwd_ex(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int R, int K,
                    int N, const feta_colsum_seg* segs, int nseg, int compute, feta_stream_t stream);

/* ---- A3: dynamic Chebyshev filter, direct recursion on a dense scaled Laplacian ----
 * Replaces ChebConvDynamic.forward + __norm__ (transformer/ChebNetDynamic.py:108-189)
 * and its head-stacking / gather / scatter glue (transformer/models.py:178-186,
 * 200-202,346-360):
 *   y[b,i,h,:] = sum_k (T_k(Lhat_b) x_bh)[i,:] . W_k^(h,b) + bias   for i < n_real[b],
 *   y = 0 on padded rows; W^(h,b) = coeff[h*B+b].reshape(P, dh, dh).
 * lhat   [B,N,N] dense operator of one propagate(): out = Lhat @ x
 * heads_share_graph  0 = reference-literal (only head 0 sees the graph, heads >= 1
 *                    use Lhat = 0: models.py:186); 1 = every head filtered
 * x, y   token tensors (x strides x_sb,x_sn ; y strides y_sb,y_sn)
 */
int feta_cheb_filter_fwd(const float* x, int64_t x_sb, int64_t x_sn,
                         const float* lhat, const float* coeff, const float* bias,
                         const int32_t* n_real,
                         float* y, int64_t y_sb, int64_t y_sn,
                         int B, int N, int H, int dh, int P, int heads_share_graph,
                         feta_stream_t stream);

/* dx (x strides), dcoeff [H*B, P*dh*dh], dbias_part [B*H, dh] (reduce with feta_colsum) */
int feta_cheb_filter_bwd(const float* x, int64_t x_sb, int64_t x_sn,
                         const float* lhat, const float* coeff, const int32_t* n_real,
                         const float* dy, int64_t y_sb, int64_t y_sn,
                         float* dx, float* dcoeff, float* dbias_part,
                         int B, int N, int H, int dh, int P, int heads_share_graph,
                         feta_stream_t stream);

/* ---- A3 (eigenbasis form): U^T.X -> g(Lambda) -> U.(.) ------------------------------
 * Same operator in the Laplacian eigenbasis:
 *   y = U [ sum_k diag(t_k(lam)) (U^T x) W_k ] + bias ,  t_0=1, t_1=lam, t_k=2 lam t_{k-1}-t_{k-2}
 * exact w.r.t. feta_cheb_filter_* iff K spans the graph (K >= n_real); K < n is the
 * truncated benchmark operator (BASELINE configs K=8/16/32).
 * u [B,N,K] (rows >= n_real zero), lam [B,K].
 */
int feta_spec_filter_fwd(const float* x, int64_t x_sb, int64_t x_sn,
                         const float* u, const float* lam,
                         const float* coeff, const float* bias, const int32_t* n_real,
                         float* y, int64_t y_sb, int64_t y_sn,
                         int B, int N, int H, int dh, int P, int K, int heads_share_graph,
                         feta_stream_t stream);

int feta_spec_filter_bwd(const float* x, int64_t x_sb, int64_t x_sn,
                         const float* u, const float* lam,
                         const float* coeff, const int32_t* n_real,
                         const float* dy, int64_t y_sb, int64_t y_sn,
                         float* dx, float* dcoeff, float* dbias_part,
                         int B, int N, int H, int dh, int P, int K, int heads_share_graph,
                         feta_stream_t stream);

/* ---- linear_cat folded into the per-graph eigenbasis filter (ABI 10) -------------------------------------------
 * transformer/models.py:223-224: output = linear_cat(cat(output, allout_filtered)).  With W_cat = [Wa | Wb] ([64][128]):
 *   out = xn Wa^T + filt Wb^T + b_cat,  filt = U Ytil + bias  =>  filt Wb^T = U (Ytil Wb^T) + bias Wb^T
 * - the filter half is a [K x 64] x [64 x 64] product in the eigen domain, the stack half the graph's own rows through a
 * 64 x 64 product: both are done by the workgroup that filters the graph (feta_spec_cat_supported: 4 heads x 16, order 4,
 * N <= 128, K <= 32, every head on the graph).  y (= filt) is still written (linear_cat's backward contracts it).
 * y2: the stack output rows [N*B][64], row(b, i) = b*y2_sb + i*y2_sn elements; seen through its last BatchNorm when
 * y2_stats (fresh partial sums [Gx + 1][2][64]: finalized here over M rows - workgroup 0 publishes bn_out [4][64] and
 * updates rmean / rvar / nbt, as feta_rowlin_fwd_ex does for its x operand) or y2_bn (a published block) is given, else
 * used as it is (LayerNorm stack).  Rows of padded nodes take part (filt is zero there).  out has y's strides. */
typedef struct feta_spec_cat {
  const float* y2;
  int64_t y2_sb, y2_sn;
  const float* y2_bn;
  const float* y2_stats;
  int Gx;
  const float* gamma;
  const float* beta;
  float* bn_out;
  float* rmean;
  float* rvar;
  int64_t* nbt;
  float momentum, eps;
  int M;
  const float* w_cat;
  const float* b_cat;
  float* out;
} feta_spec_cat;
int feta_spec_cat_supported(int N, int H, int dh, int P, int K, int heads_share_graph);
int feta_spec_filter_cat_fwd(const float* x, int64_t x_sb, int64_t x_sn, const float* u, const float* lam,
                             const float* coeff, const float* bias, const int32_t* n_real, float* y,
                             int64_t y_sb, int64_t y_sn, int B, int N, int H, int dh, int P, int K,
                             int heads_share_graph, const feta_spec_cat* cat, feta_stream_t stream);

/* Backward of the same fold (ABI 11): feta_spec_filter_bwd with the backward of linear_cat inside - replaces
 * feta_rowlin_bwd_ex over [x_n | filt] (functional.RowLinearCat[BN]Fn.backward, transformer/models.py:223-224 under
 * autograd) + feta_spec_filter_bwd.  dout [N*B rows][64] = gradient w.r.t. linear_cat's output, rows addressed with the
 * strides of y2; filt = the forward's y (strides y_sb, y_sn); y2_bn = the block the forward published (or NULL: x_n = y2).
 * Outputs: dx / dcoeff / dbias_part as feta_spec_filter_bwd; dxn [rows][64] = dout W_a (gradient w.r.t. the NORMALISED
 * stack output); gs [R][2][64] = per-workgroup (sum dxn, sum dxn * xhat) for the BatchNorm backward of the stack's last norm
 * (NULL without y2_bn); partial [R][partial_ld] = [dW_cat 64 x 128 | db_cat 64] per workgroup, reduced by the caller's
 * feta_colsum_multi.  R = feta_spec_cat_bwd_rows(B) rows in gs and partial: one per graph up to 512 graphs, beyond that a
 * workgroup walks graphs blockIdx, blockIdx + 512, ... and leaves their sum.  P = 4, 4 heads x 16, N <= 64, K <= 32, fp32, every head on the graph (feta_spec_cat_bwd_supported). */
typedef struct feta_spec_cat_grad {
  const float* dout;
  const float* y2;
  int64_t y2_sb, y2_sn;
  const float* y2_bn;
  const float* filt;
  const float* w_cat;
  float* dxn;
  float* gs;
  float* partial;
  int64_t partial_ld;
} feta_spec_cat_grad;
int feta_spec_cat_bwd_supported(int N, int H, int dh, int P, int K, int heads_share_graph);
int feta_spec_cat_bwd_rows(int B);
int feta_spec_filter_cat_bwd(const float* x, int64_t x_sb, int64_t x_sn, const float* u, const float* lam,
                             const float* coeff, const int32_t* n_real, int64_t y_sb, int64_t y_sn, float* dx,
                             float* dcoeff, float* dbias_part, int B, int N, int H, int dh, int P, int K,
                             int heads_share_graph, const feta_spec_cat_grad* cat, feta_stream_t stream);

/* ---- bf16 STORAGE variants of A1 and A3 (BASELINE configs 3 and 5) ----------------------------------
 * Same operators, same argument meaning as feta_attn_fwd/bwd and feta_spec_filter_fwd/bwd; `void*`
 * operands are bf16 (q, k, v, pe, out, attn, dout, dq, dk, dv; x, u, coeff, y, dy, dx, dcoeff), base
 * pointers 8-byte aligned, strides in elements (multiples of 4).  stats, delta, lam, bias and dbias_part
 * stay fp32; the contractions run on v_mfma_f32_16x16x16_bf16 with fp32 accumulation, softmax statistics
 * and t_k(lambda) in fp32.  The reference has no reduced-precision mode: parity of these entry points is
 * a stated tolerance against the fp64 oracle (tests/), the fp32 entry points are the reference arithmetic.
 * dh in {16, 32, 64} (attention), {16, 32} (filter); N <= FETA_MAX_NODES; K <= 64. */
int feta_attn_fwd_bf16(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                       const void* pe, const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn,
                       void* attn, float* stats, float scale, int B, int N, int H, int dh,
                       feta_stream_t stream);
int feta_attn_bwd_bf16(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                       const void* pe, const int32_t* n_real, const void* out, const void* dout,
                       int64_t o_sb, int64_t o_sn, const float* stats, float* delta,
                       void* dq, void* dk, void* dv, float scale, int B, int N, int H, int dh,
                       feta_stream_t stream);
int feta_spec_filter_fwd_bf16(const void* x, int64_t x_sb, int64_t x_sn, const void* u, const float* lam,
                              const void* coeff, const float* bias, const int32_t* n_real,
                              void* y, int64_t y_sb, int64_t y_sn,
                              int B, int N, int H, int dh, int P, int K, int heads_share_graph,
                              feta_stream_t stream);
int feta_spec_filter_bwd_bf16(const void* x, int64_t x_sb, int64_t x_sn, const void* u, const float* lam,
                              const void* coeff, const int32_t* n_real,
                              const void* dy, int64_t y_sb, int64_t y_sn,
                              void* dx, void* dcoeff, float* dbias_part,
                              int B, int N, int H, int dh, int P, int K, int heads_share_graph,
                              feta_stream_t stream);

/* ---- A1 with attention-probability dropout (step (6) of the reconstructed layer; the reference
 * scripts expose --dropout, default 0.0: experiments/run_transformer_gengcn.py:47) --------------------------
 * attn = dropout(softmax-like probabilities, p) (kept entries scaled by 1/(1-p)); out = attn . v; the written
 * attn is the dropped one (as nn.MultiheadAttention returns it).  The keep mask is a pure function of
 * (seed, offset, b, h, query, key): Philox4x32-10, key = seed, counter = (g, offset) with
 * g = ((b*H + h)*N + query) * ceil(N/4) + key/4, word key%4 of the output, keep iff word >= floor(p * 2^32);
 * backward regenerates it from the same (seed, offset) - no mask tensor.  dtype: FETA_DTYPE_F32 (float
 * operands, 16-byte aligned) or FETA_DTYPE_BF16 (as the *_bf16 entry points). */
#define FETA_DTYPE_F32 0
#define FETA_DTYPE_BF16 1
int feta_attn_fwd_drop(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                       const void* pe, const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn,
                       void* attn, float* stats, float scale, float p_drop, uint64_t seed, uint64_t offset,
                       int dtype, int B, int N, int H, int dh, feta_stream_t stream);
int feta_attn_bwd_drop(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                       const void* pe, const int32_t* n_real, const void* out, const void* dout,
                       int64_t o_sb, int64_t o_sn, const float* stats, float* delta,
                       void* dq, void* dk, void* dv, float scale, float p_drop, uint64_t seed, uint64_t offset,
                       int dtype, int B, int N, int H, int dh, feta_stream_t stream);
/* the same with the key DEVICE-RESIDENT (ABI 7): state = {seed, offset} (two uint64 in device memory), the mask of this
 * call is the one of (state[0], state[1] + offset_add), read when the kernel RUNS - a captured hipGraph (the reference's
 * --dropout > 0 inside train.GraphedTrainStep) draws fresh masks on every replay; the step advances state[1] itself. */
int feta_attn_fwd_drop_dev(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                           const void* pe, const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn,
                           void* attn, float* stats, float scale, float p_drop, const uint64_t* state,
                           uint64_t offset_add, int dtype, int B, int N, int H, int dh, feta_stream_t stream);
int feta_attn_bwd_drop_dev(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                           const void* pe, const int32_t* n_real, const void* out, const void* dout,
                           int64_t o_sb, int64_t o_sn, const float* stats, float* delta,
                           void* dq, void* dk, void* dv, float scale, float p_drop, const uint64_t* state,
                           uint64_t offset_add, int dtype, int B, int N, int H, int dh, feta_stream_t stream);

/* ---- A1 with a choice of exponent stabilisation (stab = {rowmax, clamp5}; ABI 7) --------------------
 * FETA_STAB_ROWMAX: e = exp(s - rowmax) - feta_attn_fwd / feta_attn_bwd (upstream GraphiT, README.md:129);
 * FETA_STAB_CLAMP5: e = exp(clamp(s, -5, 5)), the form of the in-tree DGL witnesses (LSPE/layers/graphit_gt_layer.py:
 * 39-43, LPE/layers/graph_transformer_spectra_layer.py:239-243): no row maximum (stats holds 0 for it), a clamped
 * score passes no gradient; everything else (key mask, * pe, / max(rowsum, 1e-6)) as feta_attn_fwd.  A1's source is
 * absent from the reference, so the open semantic choices are flags.  dtype as feta_attn_fwd_drop. */
#define FETA_STAB_ROWMAX 0
#define FETA_STAB_CLAMP5 1
int feta_attn_fwd_stab(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                       const void* pe, const int32_t* n_real, void* out, int64_t o_sb, int64_t o_sn,
                       void* attn, float* stats, float scale, int stab, int dtype, int B, int N, int H, int dh,
                       feta_stream_t stream);
int feta_attn_bwd_stab(const void* q, const void* k, const void* v, int64_t qkv_sb, int64_t qkv_sn,
                       const void* pe, const int32_t* n_real, const void* out, const void* dout,
                       int64_t o_sb, int64_t o_sn, const float* stats, float* delta,
                       void* dq, void* dk, void* dv, float scale, int stab, int dtype, int B, int N, int H, int dh,
                       feta_stream_t stream);

/* ---- A1/A4: row-wise linears of the encoder layer and BatchNorm1d --------------------------
 * Replaces the F.linear / relu / degree scaling / residual / BatchNorm1d sequence of
 * DiffTransformerEncoderLayer.forward (contract transformer/models.py:166-167; body per upstream
 * GraphiT, README.md:129) and linear_cat (transformer/models.py:223-224).  Activations are
 * [M, C] row-major, M = N*B rows.
 *   y = relu?(x W^T + bias) * rowscale[row]? + residual?        x [M,KI], W [NO,KI], y [M,NO]
 * stats (optional) receives per-block partial (sum, sumsq) of y: [feta_rowlin_blocks(M) + 1, 2, NO] (ABI 7: the last
 * row records the shift of the sums, feta_rowlin_ex.stats_shift - zero through this entry point),
 * the BatchNorm statistics of the result without a separate pass.
 * KI in {16,32,64,128,192,256}; NO a multiple of 16 (backward: NO in the same set).
 */
int feta_rowlin_blocks(int M);
int feta_rowlin_chunks(int M);
int feta_rowlin_fwd(const float* x, const float* w, const float* bias, const float* rowscale,
                    const float* residual, float* y, float* stats, int relu,
                    int M, int KI, int NO, feta_stream_t stream);
/* g = dy * rowscale? * [ysaved > 0]? ;  dx = g W ;  dwdb = [ g^T x  (NO*KI) | colsum g (NO) ].
 * partial: scratch [feta_rowlin_chunks(M), NO*KI + NO].  dX and the split-K weight gradient run
 * in one launch (blocks take roles), then one deterministic reduction of the partials. */
int feta_rowlin_bwd(const float* x, const float* w, const float* dy, const float* rowscale,
                    const float* ysaved, float* dx, float* partial, float* dwdb,
                    int M, int KI, int NO, feta_stream_t stream);

/* Extended form used by the fused encoder stack: the same two kernels with BatchNorm folded into
 * the operand loads and epilogues, so that BatchNorm never runs as a pass of its own.
 * "bn parameter block" = float[4][D]: scale = gamma*rstd, shift = beta - mean*scale, mean, rstd.
 * Unused pointers are NULL.  All tensors [M, .] row-major fp32. */
typedef struct feta_rowlin_ex {
  /* common */
  const float* x;         /* [M,KI] input (pre-norm values when x_bn / x_stats is set) */
  const float* w;         /* [NO,KI] */
  const float* bias;      /* [NO] */
  const float* rowscale;  /* [M] */
  int M, KI, NO, relu;
  /* forward:  y = relu?(BN?(x) W^T + bias) * rowscale? + BN?(residual)? */
  const float* residual;  /* [M,NO] */
  const float* res_bn;    /* bn parameter block [4][NO] through which residual is seen */
  float* y;               /* [M,NO] */
  float* stats;           /* [feta_rowlin_blocks(M) + 1,2,NO] partial (shifted) (sum, sumsq) of y + the shift row */
  const float* x_bn;      /* finalized bn parameter block [4][KI] of the input, or */
  const float* x_stats;   /* [Gx + 1,2,KI] partial statistics (+ shift row) to finalize here: needs x_gamma, x_beta, */
  int Gx;                 /*   x_bn_out (written by block 0), optional running stats */
  const float* x_gamma;
  const float* x_beta;
  float* x_bn_out;
  float* x_rmean;
  float* x_rvar;
  int64_t* x_nbt;         /* num_batches_tracked of that BatchNorm (+1 by block 0), or NULL */
  float momentum, eps;
  /* backward:  g = BNbwd?(dy) * rowscale? * [relu_y > 0]? ;  dx = g W (+ adds) ;  dW = g^T BN?(x) */
  const float* dy;        /* [M,NO] */
  const float* relu_y;    /* [M,NO] saved relu output */
  float* dx;              /* [M,KI] */
  float* partial;         /* [feta_rowlin_chunks(M), NO*KI + NO] scratch */
  int partial_ld;         /* row pitch of partial (0 = NO*KI+NO): lets several linears share one
                             [chunks, total] buffer that the caller reduces with ONE feta_colsum */
  const float* g_y;       /* [M,NO] pre-norm values of the BatchNorm whose backward is applied to dy */
  const float* g_bn;      /* its bn parameter block [4][NO] */
  const float* g_sum;     /* [Gs,2,NO] partial (sum dy, sum dy*xhat): finalized here, or */
  int Gs;
  const float* g_fin;     /* [2][NO] already finalized (mean dy, mean dy*xhat) */
  float* g_fin_out;       /* [2][NO] written by block 0 when g_sum is given */
  float* dgamma;          /* [NO] = sum dy*xhat   (written with g_sum) */
  float* dbeta;           /* [NO] = sum dy */
  const float* add_plain; /* [M,KI] added to dx */
  const float* add_dout;  /* [M,KI]: dx += BNbwd(add_dout) with add_y, add_bn [4][KI], add_fin [2][KI] */
  const float* add_y;
  const float* add_bn;
  const float* add_fin;
  const float* x2;        /* x is the virtual concatenation [x (x_split cols) | x2 (KI - x_split cols)]: */
  int x_split;            /*   linear_cat without materialising torch.cat; backward writes dx | dx2.  With x2,
                             x_bn / x_stats (and x_gamma ..) and sum_y / sum_bn / sum_out describe the BatchNorm
                             of the x part only: x_split columns instead of KI */
  float* dx2;
  const float* sum_y;     /* [M,KI] pre-norm values of the BatchNorm that produced x: emit */
  const float* sum_bn;    /*   sum_out [feta_rowlin_blocks(M),2,KI] = partial (sum dx, sum dx*xhat) */
  float* sum_out;
  const float* stats_shift; /* nullable [NO] (ABI 7): `stats` are SHIFTED sums - sum (y - K), sum (y - K)^2 with K = this
                               vector, the running mean of the BatchNorm that will consume them (0 if NULL) - and K is
                               recorded in one extra row: stats / x_stats / y_stats buffers hold G + 1 rows, row G =
                               [K | unused].  mean = K + S1 / M, var = S2 / M - (S1 / M)^2: the cancellation of
                               E[y^2] - mean^2 moves from |mean| to |mean - running mean| (csrc/feta_rowops.h) */
} feta_rowlin_ex;

int feta_rowlin_fwd_ex(const feta_rowlin_ex* d, feta_stream_t stream);
/* dwdb [NO*KI + NO] receives the reduced weight and bias gradient (NULL with partial_ld > 0:
 * partials only). */
int feta_rowlin_bwd_ex(const feta_rowlin_ex* d, float* dwdb, feta_stream_t stream);

/* out = BN(y) written explicitly (end of the stack), publishing the bn parameter block [4][D].
 * stats holds G_stats partial rows and, behind them, the shift row: [G_stats + 1][2][D] (G_stats = 0 means
 * feta_rowlin_blocks(M), the producer was a feta_rowlin_fwd*; feta_ffn_blocks(M) after feta_ffn_fwd;
 * feta_attn_block_stat_rows(B, N) after feta_attn_block_fwd). */
int feta_bn_apply_fwd_prm(const float* y, const float* stats, const float* gamma, const float* beta,
                          float* out, float* bn_prm, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float momentum, float eps, int M, int D, int G_stats,
                          feta_stream_t stream);
/* partial [feta_rowlin_blocks(M),2,D] = per-block (sum dout, sum dout*xhat), bn_prm [4][D]. */
int feta_bn_bwd_reduce(const float* y, const float* dout, const float* bn_prm, float* partial,
                       int M, int D, feta_stream_t stream);

/* Training-mode BatchNorm1d over the M rows (padded rows included, as nn.BatchNorm1d on the
 * [N*B, d] view does).  stats [feta_rowlin_blocks(M) + 1, 2, D] from feta_rowlin_fwd or feta_bn_stats:
 * G = feta_rowlin_blocks(M) partial rows PLUS ONE shift row (ABI 7: the sums are relative to the shift recorded in
 * row G; feta_bn_stats writes zeros there) - a caller that sizes the buffer with G rows gets a D-float overrun.
 * mean_rstd [2, D] is saved for backward; running_* (nullable) are updated with momentum and the
 * unbiased variance, num_batches_tracked (nullable, int64) is advanced by one, as nn.BatchNorm1d does. */
int feta_bn_stats(const float* y, float* stats, int M, int D, feta_stream_t stream);
/* ... with a shift (feta_rowlin_ex.stats_shift); either way stats has feta_rowlin_blocks(M) + 1 rows (ABI 7) */
int feta_bn_stats_shift(const float* y, const float* shift, float* stats, int M, int D, feta_stream_t stream);
int feta_bn_apply_fwd(const float* y, const float* stats, const float* gamma, const float* beta,
                      float* out, float* mean_rstd, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float momentum, float eps, int M, int D, feta_stream_t stream);
/* partial: scratch [feta_rowlin_blocks(M), 2, D]; dgamma, dbeta [D]. */
int feta_bn_bwd(const float* y, const float* dout, const float* mean_rstd, const float* gamma,
                float* partial, float* dy, float* dgamma, float* dbeta,
                int M, int D, feta_stream_t stream);

/* ---- attention sub-block of one encoder layer in ONE launch --------------------------------
 * Replaces, for d_model = 64 = 4 heads x 16 and N <= 64 (feta_attn_block_supported), the sequence
 * feta_rowlin_fwd_ex (in_proj) -> feta_attn_fwd -> feta_rowlin_fwd_ex (out_proj + degree + residual
 * + BatchNorm statistics) of DiffTransformerEncoderLayer.forward (contract
 * transformer/models.py:166-167,179,244; body per upstream GraphiT, README.md:129).
 * One workgroup per graph; rows of the [M = B*N, .] activations are addressed as
 * row(b, i) = b*row_sb + i*row_sn (seq-first [N,B,d]: row_sb = 1, row_sn = B).
 * x is "seen through a BatchNorm" exactly as in feta_rowlin_ex: either x_bn (a published [4][64]
 * parameter block) or x_stats [Gx][2][64] fresh partial sums that this launch finalizes
 * (publishing x_bn_out and updating x_rmean / x_rvar), or neither (first layer).
 * Outputs: qkv [M,192] (in_proj result, for backward), out [M,64] (per-head attention outputs,
 * concatenated: out_each_head), attn_stats [B,4,N,2] (row max, un-clamped row sum), attn
 * [B,4,N,N] or NULL, y [M,64] = x_norm + rowscale * (out W_out^T + b_out), y_stats [G][2][64]
 * partial (sum, sum of squares) over the rows of y - padded rows included, as nn.BatchNorm1d over the
 * [N*B, d] view counts them; NULL (ABI 9): no statistics are taken (LayerNorm stack).
 * G = feta_attn_block_stat_rows(B, N) (ABI 8): one row per WORKGROUP - a graph is one
 * workgroup of eight waves (head x query-tile parity), or two such workgroups (query tiles split between them, K and V
 * projected by both) where a graph has more than two 16-row tiles and 2 B workgroups still fit the chip; the shift row
 * (y_shift) is row G. */
typedef struct feta_attn_block {
  const float* x;
  const float* x_bn;
  const float* x_stats;
  int Gx;
  const float* x_gamma;
  const float* x_beta;
  float* x_bn_out;
  float* x_rmean;
  float* x_rvar;
  int64_t* x_nbt;      /* num_batches_tracked of that BatchNorm (+1 by workgroup 0), or NULL */
  float momentum, eps;
  const float* w_in;   /* [192,64] */
  const float* b_in;   /* [192] or NULL */
  const float* w_out;  /* [64,64] */
  const float* b_out;  /* [64] or NULL */
  const float* pe;     /* [B,N,N] or NULL */
  const int32_t* n_real;
  const float* rowscale; /* [M] degree scale per row, or NULL */
  float* qkv;
  float* out;
  float* attn_stats;
  float* attn;
  float* y;
  float* y_stats;
  float scale;         /* d_h^-1/2 */
  int B, N, M;
  int64_t row_sb, row_sn;
  int tie_qk;
  int dtype;           /* FETA_F32 | FETA_BF16: storage type of x, pe, qkv, out, y [T] (ABI 7) */
  const float* y_shift; /* nullable [64]: shift of the y statistics = running mean of the BatchNorm that will normalise y
                           (feta_rowlin_ex.stats_shift); y_stats has G + 1 rows (G = feta_attn_block_stat_rows), row G
                           receives the shift */
  float* out_f32;      /* nullable: `out` once more, as fp32 [M,64] - the fp32 filter stage behind a bf16 stack reads it
                          (out_each_head of the last layer, transformer/models.py:179) without a cast launch */
  const float* x_ln_gamma; /* nullable [64] (ABI 9): x is seen through a LAYERNORM - the rows of x are the PRE-norm rows
                              (the previous layer's y2) and x_norm = (x - mean_row) * rstd_row * gamma + beta is computed
                              per row when the row is staged (eps: the field above; biased variance, as F.layer_norm):
                              LayerNorm is row-local, so its consumer applies it and no normalised tensor is ever
                              written (norm2 of DiffTransformerEncoderLayer with batch_norm=False, the reference's
                              default: experiments/run_transformer_gengcn_cv.py:56).  Excludes x_bn / x_stats. */
  const float* x_ln_beta;  /* [64], with x_ln_gamma */
} feta_attn_block;

int feta_attn_block_supported(int N, int d_model, int heads);
int feta_attn_block_stat_rows(int B, int N);   /* partial rows a forward launch writes into y_stats (0: unsupported shape) */
int feta_attn_block_fwd(const feta_attn_block* d, feta_stream_t stream);
/* the same launch with up to FETA_COLSUM_MAX_SEGS independent column sums in trailing workgroups (ABI 6): at the
 * BASELINE batch the first launch of a forward pass leaves half the chip idle, and s = colsum(gcn.weight) of the
 * coefficient generator (transformer/models.py:280-282: the GCN only ever sees an all-ones input) depends on the
 * parameters alone. */
int feta_attn_block_fwd_sums(const feta_attn_block* d, const feta_colsum_seg* segs, int nseg, feta_stream_t stream);

/* ---- attention core + out_proj behind the in_proj launch, for graphs beyond feta_attn_block (ABI 8) -------------
 * (feta_attn_out_supported: d_model = 64, 4 heads, N <= 256.)  Replaces feta_attn_fwd -> feta_rowlin_fwd_ex
 * (out_proj + degree + residual + BatchNorm statistics) of DiffTransformerEncoderLayer.forward for config 4 (PATTERN,
 * experiments/run_transformer_gengcn_SBM_cv.py; contract transformer/models.py:166-167,179,244): one workgroup per
 * (graph, 32 query rows), the per-head outputs meet in LDS and are the out_proj operand.  Same descriptor as
 * feta_attn_block, read as follows: qkv [M,192] is an INPUT (the in_proj result), x is the residual (seen through the
 * published parameter block x_bn when given; x_stats must be NULL - the in_proj launch finalized them), w_in / b_in are
 * ignored; outputs out, out_f32, attn_stats, attn, y as there; y_stats (nullable) [G + 1][2][64] with
 * G = feta_attn_out_stat_rows(B, N) = B * ceil(N / 32) partial rows and the shift row behind them.  fp32 token tensors
 * (dtype = FETA_F32) only. */
int feta_attn_out_supported(int N, int d_model, int heads);
int feta_attn_out_stat_rows(int B, int N);
int feta_attn_out_fwd(const feta_attn_block* d, feta_stream_t stream);
/* ... with column sums in trailing workgroups, as feta_attn_block_fwd_sums (s = colsum(gcn.weight) in the first launch) */
int feta_attn_out_fwd_sums(const feta_attn_block* d, const feta_colsum_seg* segs, int nseg, feta_stream_t stream);

/* ---- backward of the attention sub-block in ONE launch ------------------------------------------------------
 * (feta_attn_block_bwd_supported: d_model = 64, 4 heads, N <= 64; K not tied to Q.)  Replaces
 * feta_rowlin_bwd_ex (out_proj) -> feta_attn_bwd -> feta_rowlin_bwd_ex (in_proj) of the layer's backward:
 *   g1 = BatchNorm-1 backward of dy (y1 [M,64], bn1 [4][64], partial sums g_sum [Gs][2][64] finalized here ->
 *        fin_out [2][64], dgamma, dbeta) - or dy itself when y1 is NULL (LayerNorm stack);
 *   dconcat = (rowscale * g1) W_out (+ dout2 [M,64], the filter branch's gradient into out_each_head);
 *   dq|dk|dv from q|k|v (qkv [M,192]), out [M,64], pe, attn_stats (feta_attn_bwd's arithmetic);
 *   dx [M,64] = dqkv W_in + g1;  sum_out (nullable) [2 * feta_attn_block_bwd_blocks(B)][2][64]: partial
 *        (sum dx, sum dx * xhat0), xhat0 from x0 (pre-norm) and bn0 [4][64];
 *   partial: one row per workgroup (feta_attn_block_bwd_blocks(B) rows: one workgroup per graph up to 256 graphs, beyond
 *        that the workgroups walk several graphs and add into their row; pitch partial_ld, 0: 4*64*64 + 4*64),
 *        columns [dW_out (64 x 64) | db_out (64) | dW_in (192 x 64) | db_in (192)], reduced by the caller;
 *        dW_in contracts dqkv with x0 seen through bn0 (scale, shift rows) when given.
 * Rows are addressed as in feta_attn_block: row(b, i) = b*row_sb + i*row_sn. */
typedef struct feta_attn_block_grad {
  const float* dy;
  const float* y1;
  const float* bn1;
  const float* g_sum;
  int Gs;
  float* fin_out;
  float* dgamma;
  float* dbeta;
  const float* rowscale;
  const float* w_out;
  const float* w_in;
  const float* qkv;
  const float* out;
  const float* dout2;
  const float* pe;
  const int32_t* n_real;
  const float* attn_stats;
  const float* x0;
  const float* bn0;
  float* dx;
  float* dx_b;       /* nullable: SPLIT form, two workgroups per graph (one per pair of heads): dx receives pair 0's part
                        (with the residual g1), dx_b pair 1's; the consumer adds them (feta_ffn_grad.dy_b); sum_out
                        rows then are partial sums of the two parts - same row count */
  float* sum_out;
  float* partial;
  int partial_ld;
  float scale;
  int B, N, M;
  int64_t row_sb, row_sn;
  int dtype;         /* FETA_F32 | FETA_BF16: storage type of dy, y1, qkv, out, dout2, pe, x0, dx, dx_b [T] (ABI 7) */
  int dout2_f32;     /* 1: dout2 is fp32 whatever dtype says (it comes from the fp32 filter stage) */
  const float* ln1_gamma;   /* nullable [64] (ABI 9): dy is the gradient w.r.t. LN1(y1) = norm1's output, and
                               g1 = LayerNorm backward of dy, computed per row when the gradient row is staged:
                               xhat = (y1 - mean_row) rstd_row, g = dy * gamma, g1 = rstd_row (g - mean_row(g) -
                               xhat mean_row(g xhat)); y1 = the pre-norm rows (bn1 / g_sum must be NULL).  The partial
                               row gains [dgamma1 (64) | dbeta1 (64)] = per-workgroup sums of (dy xhat, dy) behind
                               db_in (default pitch 4*64*64 + 4*64 + 128). */
  const float* x0_ln_gamma; /* nullable [64]: x0 holds PRE-norm rows, the in_proj operand is LayerNorm(x0) * gamma +
                               beta computed per row on load (feta_attn_block.x_ln_gamma); excludes bn0 */
  const float* x0_ln_beta;
  float ln_eps;
} feta_attn_block_grad;

int feta_attn_block_bwd_supported(int N, int d_model, int heads);
int feta_attn_block_bwd_blocks(int B);
int feta_attn_block_bwd(const feta_attn_block_grad* d, feta_stream_t stream);
/* the same launch with up to FETA_COLSUM_MAX_SEGS independent column sums in trailing workgroups (ABI 8): the last launch
 * of a stack's backward leaves half the chip idle at the BASELINE batch while the split-K partials of everything behind it
 * are complete - reduced here, the final reduction launch is left with this launch's own partial columns. */
int feta_attn_block_bwd_sums(const feta_attn_block_grad* d, const feta_colsum_seg* segs, int nseg, feta_stream_t stream);

/* ---- feed-forward half of one encoder layer in ONE launch -----------------------------------
 * x = BN1(y1) (x_bn | x_stats as in feta_rowlin_ex / feta_attn_block);  h = relu(x W1^T + b1);
 * y = x + h W2^T + b2;  y_stats [feta_ffn_blocks(M)][2][64] per-workgroup (sum, sum of squares).
 * Replaces the two feta_rowlin_fwd_ex launches of linear1 / linear2 for d_model = 64 and
 * dim_feedforward in {64,128,256} (feta_ffn_supported).  h [M,FF] is written for backward. */
typedef struct feta_ffn {
  const float* x;
  const float* x_bn;
  const float* x_stats;
  int Gx;
  const float* x_gamma;
  const float* x_beta;
  float* x_bn_out;
  float* x_rmean;
  float* x_rvar;
  int64_t* x_nbt;
  float momentum, eps;
  const float* w1;  /* [FF,64] */
  const float* b1;  /* [FF] or NULL */
  const float* w2;  /* [64,FF] */
  const float* b2;  /* [64] or NULL */
  float* h;
  float* y;
  float* y_stats;   /* or NULL */
  int M, FF;
  int dtype;        /* FETA_F32 | FETA_BF16: storage type of x, h, y [T] (ABI 7) */
  const float* y_shift; /* nullable [64]: shift of the y statistics (feta_attn_block.y_shift); y_stats has
                           feta_ffn_blocks(M) + 1 rows */
  int y_f32;        /* 1: y is written as fp32 whatever dtype says (last layer of a bf16 stack: its consumer, linear_cat
                       with the folded BatchNorm, transformer/models.py:223-224, is an fp32 kernel) */
  const float* x_ln_gamma; /* nullable [64] (ABI 9): x = LayerNorm(x rows) * gamma + beta, computed per row on load
                              (feta_attn_block.x_ln_gamma: norm1 with batch_norm=False); excludes x_bn / x_stats */
  const float* x_ln_beta;
  float* y_ln_out;          /* nullable [M,64] [T, or fp32 with y_ln_f32] (ABI 11): LayerNorm(y rows) * y_ln_gamma + y_ln_beta, */
  const float* y_ln_gamma;  /*   written beside y by the epilogue - norm2 of a LayerNorm stack where its consumer is not an */
  const float* y_ln_beta;   /*   on-load kernel (graphs beyond 64 nodes; the end of the stack).  Excludes y_stats. */
  float y_ln_eps;
  int y_ln_f32;
} feta_ffn;

int feta_ffn_supported(int d_model, int ff);
int feta_ffn_blocks(int M);
int feta_ffn_fwd(const feta_ffn* d, feta_stream_t stream);
/* the same launch with the FORWARD of the coefficient generator (feta_coeff_fwd's arguments) in trailing workgroups
 * (ABI 6): get_filter_coefficients (transformer/models.py:240-283) needs only the attention matrix of the last layer,
 * which the launch before this one wrote - two half-empty launches of a captured step become one. */
typedef struct feta_coeff_fwd_role {
  const float* attn; const int32_t* n_real; const float* s; const float* gcn_bias; float* cj; float* pooled;
  int B, N, H, C;
} feta_coeff_fwd_role;
int feta_ffn_fwd_coeff(const feta_ffn* d, const feta_coeff_fwd_role* c, feta_stream_t stream);

/* ---- backward of the feed-forward half in ONE launch ------------------------------------------------
 * (feta_ffn_bwd_supported: d_model = 64, dim_feedforward in {64,128}.)  Replaces the two feta_rowlin_bwd_ex
 * launches of linear2 and linear1 of DiffTransformerEncoderLayer (contract transformer/models.py:166-167;
 * y2 = x1 + linear2(relu(linear1(x1)))):
 *   g2 = BatchNorm-2 backward of dy (g_y = y2 [M,64], g_bn [4][64], partial sums g_sum [Gs][2][64] finalized
 *        here -> g_fin_out [2][64], dgamma, dbeta; or g_fin already finalized) - or dy itself when g_y is NULL
 *        (LayerNorm stack: dy is the gradient w.r.t. y2);
 *   dh = (g2 W2) * [h > 0] (never written to memory);  dx = g2 + dh W1;
 *   sum_out (nullable) [feta_ffn_bwd_blocks(M)][2][64]: partial (sum dx, sum dx * xhat) with xhat from x = y1
 *        (pre-norm) and x_bn [4][64], for the BatchNorm-1 backward;
 *   partial: one row per chunk - feta_ffn_bwd_chunks(M, FF) rows (ABI 11; feta_rowlin_chunks(M) or half of it) -, pitch partial_ld (0: 2*64*FF + 64 + FF), columns
 *        [dW2 (64 x FF) | db2 (64) | dW1 (FF x 64) | db1 (FF)], reduced by the caller (feta_colsum).
 * x is seen through x_bn (scale, shift rows) when given, else used as it is. */
typedef struct feta_ffn_grad {
  const float* dy;
  const float* dy_b;   /* nullable: second part of the gradient, added to dy on load (feta_attn_block_grad.dx_b) */
  const float* g_y;
  const float* g_bn;
  const float* g_sum;
  int Gs;
  const float* g_fin;
  float* g_fin_out;
  float* dgamma;
  float* dbeta;
  const float* h;    /* [M,FF] saved relu output */
  const float* w2;   /* [64,FF] */
  const float* w1;   /* [FF,64] */
  const float* x;    /* [M,64] */
  const float* x_bn; /* [4][64] or NULL */
  float* dx;         /* [M,64] */
  float* sum_out;
  float* partial;
  int partial_ld;
  int M, FF;
  int dtype;         /* FETA_F32 | FETA_BF16: storage type of dy, dy_b, g_y, h, x, dx [T] (ABI 7) */
  int g_f32;         /* 1: dy and g_y are fp32 whatever dtype says (last layer of a bf16 stack, see feta_ffn.y_f32) */
  const float* g_ln_gamma; /* nullable [64] (ABI 9): dy is the gradient w.r.t. LN2(y2) = norm2's output and
                              g2 = LayerNorm backward of dy per row, computed on load from g_y = y2 (pre-norm rows; g_bn /
                              g_sum / g_fin must be NULL), as feta_attn_block_grad.ln1_gamma.  The partial rows gain
                              [dgamma2 (64) | dbeta2 (64)] behind db1 (default pitch 2*64*FF + 64 + FF + 128). */
  const float* x_ln_gamma; /* nullable [64]: x holds PRE-norm rows (y1), the operand is LayerNorm(x) * gamma + beta per
                              row on load; excludes x_bn */
  const float* x_ln_beta;
  float ln_eps;
} feta_ffn_grad;

int feta_ffn_bwd_supported(int d_model, int ff);
int feta_ffn_bwd_blocks(int M);
int feta_ffn_bwd_chunks(int M, int ff);   /* split-K chunks = rows of `partial` a launch writes */
int feta_ffn_bwd(const feta_ffn_grad* d, feta_stream_t stream);
/* ... and the BACKWARD kernel of the coefficient generator (feta_coeff_bwd with ds = NULL: partial [G, 2, C] is left
 * for the caller's feta_colsum_multi) in trailing workgroups of the first launch of the layer stack's backward. */
typedef struct feta_coeff_bwd_role {
  const float* cj; const int32_t* n_real; const float* s; const float* gcn_bias; const float* dpooled; float* partial;
  int B, N, H, C;
} feta_coeff_bwd_role;
int feta_ffn_bwd_coeff(const feta_ffn_grad* d, const feta_coeff_bwd_role* c, feta_stream_t stream);

/* ---- graph preprocessing -------------------------------------------------------------
 * Dense Lhat = -D^-1/2 A D^-1/2 per graph from the batched edge list, with the exact
 * edge-list semantics of ChebConvDynamic.__norm__ (transformer/ChebNetDynamic.py:108-130):
 * self loops removed, degree scattered on the source row, duplicates summed, flow
 * source -> target (Lhat[b, t, s] += w).  lhat [B,N,N] must be zeroed by the caller;
 * deg [n_tot] is zero-initialised scratch.
 * edge_index [2,E] int64 (global node ids), node_graph [n_tot] int64, node_off [B] int32
 * (first global node id of graph b).
 */
int feta_lhat_from_edges(const int64_t* edge_index, int64_t E,
                         const int64_t* node_graph, const int32_t* node_off,
                         float* deg, float* lhat, int B, int N, int64_t n_tot,
                         feta_stream_t stream);

/* ---- LayerNorm over the feature dimension (norm1 / norm2 of DiffTransformerEncoderLayer with
 * batch_norm=False, contract transformer/models.py:505-506; the default of the TU / molhiv / SBM
 * scripts, experiments/run_transformer_gengcn_cv.py:56) --------------------------------------
 * y, out, dout, dy [M,D] row-major, D a multiple of 4, <= 256; stats [M,2] = (mean, rstd) per row
 * (biased variance, eps inside the root, as torch.nn.LayerNorm).  Backward also returns
 * dgamma_dbeta [2,D]; partial [feta_layernorm_blocks(M), 2, D] is caller-provided scratch.
 * partial_ld > 0: row pitch of partial, so that several LayerNorms share one
 * [feta_layernorm_blocks(M), total] buffer which the caller reduces with ONE feta_colsum
 * (dgamma_dbeta = NULL then), as feta_rowlin_ex.partial_ld does for the weight gradients.
 */
int feta_layernorm_blocks(int M);
int feta_layernorm_fwd(const float* y, const float* gamma, const float* beta, float eps, float* out,
                       float* stats, int M, int D, feta_stream_t stream);
int feta_layernorm_bwd(const float* dout, const float* y, const float* stats, const float* gamma,
                       float* dy, float* partial, int partial_ld, float* dgamma_dbeta, int M, int D,
                       feta_stream_t stream);
/* the same kernels with a storage type per tensor (FETA_F32 | FETA_BF16; ABI 7): the LayerNorm stack on bf16 storage
 * normalises bf16 rows into bf16 rows (fp32 into the filter stage behind the last layer), statistics and the
 * dgamma / dbeta partials stay fp32. */
int feta_layernorm_fwd_ex(const void* y, const float* gamma, const float* beta, float eps, void* out,
                          float* stats, int M, int D, int y_dtype, int out_dtype, feta_stream_t stream);
int feta_layernorm_bwd_ex(const void* dout, const void* y, const float* stats, const float* gamma,
                          void* dy, float* partial, int partial_ld, float* dgamma_dbeta, int M, int D,
                          int dout_dtype, int y_dtype, int dy_dtype, feta_stream_t stream);
/* ... with `stats` nullable (ABI 9): the forward applied this LayerNorm on load (feta_ffn.x_ln_gamma) and saved no
 * statistics - mean / rstd are recomputed from the pre-norm rows y with `eps` (the row is in registers: two more sums). */
int feta_layernorm_bwd_eps(const void* dout, const void* y, const float* stats, float eps, const float* gamma,
                           void* dy, float* partial, int partial_ld, float* dgamma_dbeta, int M, int D,
                           int dout_dtype, int y_dtype, int dy_dtype, feta_stream_t stream);

/* ---- spectrum producer ------------------------------------------------
 * Batched symmetric eigendecomposition, one workgroup per graph, the matrix in LDS (N <= 192) or, for
 * 192 < N <= 256, in a caller-provided workspace of feta_eigh_sym_workspace_bytes(B, N) bytes (0 for N <= 192;
 * the largest ogbg-molhiv bucket, BASELINE config 5, has 222 nodes):
 *   a [B,N,N]: the real n_b x n_b block of graph b is decomposed; as numpy.linalg.eigh does, only
 *   the LOWER triangle is read.  a + shift*I must be positive definite (Lhat = -D^-1/2 A D^-1/2 has
 *   its spectrum in [-1,1]: shift 2; L_sym: shift 1) - the one-sided Jacobi iteration runs on that
 *   matrix and the shift is taken off the eigenvalues again.
 *   u [B,N,K]: eigenvectors of the K smallest eigenvalues as columns, ascending; rows >= n_b and
 *   columns >= n_b are zero; the entry of largest magnitude of every column is positive.
 *   lam [B,K]: eigenvalues (0 for columns >= n_b).  sweeps [B] (may be NULL): Jacobi sweeps used.
 *   max_sweeps <= 0: 16.  tol <= 0: 1e-6 (a pair is rotated while |g_p.g_q| > tol |g_p||g_q|).
 * Replaces the per-graph host eigendecomposition of transformer/position_encoding.py:127-161 and
 * feeds feta_spec_filter_fwd/bwd (u, lam) directly.
 */
int feta_eigh_sym_supported(int N);
int64_t feta_eigh_sym_workspace_bytes(int B, int N);
int feta_eigh_sym(const float* a, const int32_t* n_real, float shift, float* u, float* lam,
                  int32_t* sweeps, float* workspace, int B, int N, int K, int max_sweeps, float tol,
                  feta_stream_t stream);

/* Kernel function of the spectrum on the real block, zero elsewhere:
 *   out_b = U_b f(lam_b + lam_offset) U_b^T,   out [B,N,N], u [B,N,K], lam [B,K]
 *   FETA_SPECTRAL_DIFFUSION: f(x) = exp(-beta x)     (DiffusionEncoding, position_encoding.py:65-72)
 *   FETA_SPECTRAL_PSTEP:     f(x) = (1 - beta x)^max(p,1)   (PStepRWEncoding, :83-93: p - 1 products)
 * zero_diag: the diagonal is cleared (PositionEncoding.apply_to, position_encoding.py:25-27).
 * With (u, lam) of Lhat pass lam_offset = 1 for the kernels of L_sym = I + Lhat.
 */
#define FETA_SPECTRAL_DIFFUSION 0
#define FETA_SPECTRAL_PSTEP 1
int feta_spectral_kernel(const float* u, const float* lam, const int32_t* n_real, int mode,
                         float beta, int p, float lam_offset, int zero_diag, float* out,
                         int B, int N, int K, feta_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FETA_HIP_H_ */
