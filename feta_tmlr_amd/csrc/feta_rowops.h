// Shared pieces of the row-wise kernels (rowwise.hip) and the fused attention block (block.hip):
// the workgroup shape and the deterministic reduction of per-block BatchNorm partial sums.
#pragma once
#include "feta_tiles.h"

// In-kernel phase stamps, diagnostic build only (-DFETA_TIMING, tools/block_timing.py): s_memtime of
// one chosen thread into a per-source-file device array.
#ifdef FETA_TIMING
#define FETA_STAMP_TO(arr, i, cond)                                                     \
  do {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    unsigned long long t_;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    if (cond) arr[i] = t_;                                                              \
  } while (0)
// s_memrealtime (100 MHz, one clock for the whole chip) of wave 0 of every 32nd workgroup into
// arr[launch & 3][workgroup / 32][stamp], over four consecutive launches (the last workgroup of a launch bumps the
// counter): start skew, phase times and the gap between launches
#define FETA_RT_STAMP(arr, launch, i)                                                                   \
  do {                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    unsigned long long t_;                                                                              \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (threadIdx.x == 0 && (blockIdx.x & 31) == 0 && blockIdx.x < 256)                                 \
      arr[(((launch) & 3) * 8 + (blockIdx.x >> 5)) * 8 + (i)] = t_;                                     \
  } while (0)
#define FETA_RT_LAUNCH_DONE(launch)                                                                     \
  do {                                                                                                  \
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) atomicAdd(&(launch), 1u);                      \
  } while (0)
#else
#define FETA_STAMP_TO(arr, i, cond)
#define FETA_RT_STAMP(arr, launch, i)
#define FETA_RT_LAUNCH_DONE(launch)
#endif

namespace feta {

constexpr int kRowWaves = 4;
constexpr int kRowThreads = 64 * kRowWaves;

// Global -> LDS staging of `count` 16-byte items by the whole workgroup: a plain `for (idx...) lds[..] = gmem[..]` loop compiles to load / wait / store per
// iteration, i.e. one full memory latency per 256 items.  src(idx) -> const float4*, dst(idx) -> float4*.
// Four requests are in flight per thread.
template <class Src, class Dst>
__device__ __forceinline__ void stage_float4(int count, Src src, Dst dst) {
  for (int base = threadIdx.x; base < count; base += kRowThreads * 4) {
    const int i0 = base, i1 = base + kRowThreads, i2 = base + 2 * kRowThreads, i3 = base + 3 * kRowThreads;
    const int last = count - 1;
    const float4 v0 = *src(i0);
    const float4 v1 = *src(i1 < count ? i1 : last);
    const float4 v2 = *src(i2 < count ? i2 : last);
    const float4 v3 = *src(i3 < count ? i3 : last);
    *dst(i0) = v0;
    if (i1 < count) *dst(i1) = v1;
    if (i2 < count) *dst(i2) = v2;
    if (i3 < count) *dst(i3) = v3;
  }
}

// Sum of the G partial pairs [G][2][D] with the first 256 threads; on return tot[c], tot[D + c] (LDS) hold the
// totals.  red: [slices][2][D] scratch.
// One partial = 2D contiguous floats = nq float4; thread -> (float4 column q, row slice).  The FIRST batch of rows
// (kPartialFirst per slice: 128 rows at D = 64 - every partial row of the BASELINE batch) can be requested ahead of
// everything else a kernel loads (partials_request) and summed later (reduce_partials_finish): after a kernel boundary
// every first touch is a ~1-2 us round trip to the memory side, and the consumers of fresh BatchNorm statistics used
// to pay one for their weights, one or two for the partial rows and one for their row tile, one after the other.
// Further rows are taken kPartialUnroll at a time (independent accumulators), the tail as one clamped batch.
#ifndef FETA_PARTIAL_UNROLL
#define FETA_PARTIAL_UNROLL 8
#endif
constexpr int kPartialUnroll = FETA_PARTIAL_UNROLL;   // 16-byte loads in flight per thread beyond the first batch
constexpr int kPartialFirst = 16;

// THREADS: threads of the workgroup that take part (all of them call; the first slices * nq do the work); NF: rows per slice
// requested in the FIRST batch.  Every partial row of the BASELINE batch must be in that batch - a row beyond it costs a
// second (and third) dependent round trip in the middle of a prologue: round 3 measured + 1.7 us on ffn_fwd when its
// producer began to emit 256 rows instead of 128 (two workgroups per graph), and the 148 rows of the feed-forward kernels
// had always cost their 512-thread consumers a second trip for the last 20.  256 threads x NF = 32 and 512 threads x
// NF = 16 both cover 256 rows.
template <int NF>
struct PartialBatchT {
  float4 v[NF];
};
typedef PartialBatchT<kPartialFirst> PartialBatch;

template <int THREADS, int NF>
__device__ __forceinline__ void partials_request_t(const float* part, int G, int D, PartialBatchT<NF>& pb) {
  const int nq = 2 * D / 4;
  const int slices = THREADS / nq > 0 ? THREADS / nq : 1;
  const int q = threadIdx.x % nq, slice = threadIdx.x / nq;
  const float4* p4 = reinterpret_cast<const float4*>(part);
  const bool mine = (int)threadIdx.x < slices * nq;
#pragma unroll
  for (int u = 0; u < NF; ++u) {
    const int gr = slice + u * slices;
    pb.v[u] = p4[(int64_t)((mine && gr < G) ? gr : 0) * nq + (mine ? q : 0)];
  }
}

template <int THREADS, int NF>
__device__ __forceinline__ void reduce_partials_finish_t(const float* part, int G, int D, const PartialBatchT<NF>& pb,
                                                         float* red, float* tot) {
  const int nq = 2 * D / 4;
  const int slices = THREADS / nq > 0 ? THREADS / nq : 1;
  const int q = threadIdx.x % nq, slice = threadIdx.x / nq;
  if ((int)threadIdx.x < slices * nq) {
    const float4* p4 = reinterpret_cast<const float4*>(part);
    constexpr int U = kPartialUnroll;
    float4 s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const float m = slice + u * slices < G ? 1.0f : 0.0f;
      s[u & 3].x += m * pb.v[u].x; s[u & 3].y += m * pb.v[u].y; s[u & 3].z += m * pb.v[u].z; s[u & 3].w += m * pb.v[u].w;
    }
    int gi = slice + NF * slices;
    for (; gi + (U - 1) * slices < G; gi += U * slices) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p4[(int64_t)(gi + u * slices) * nq + q];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[u & 3].x += v[u].x; s[u & 3].y += v[u].y; s[u & 3].z += v[u].z; s[u & 3].w += v[u].w;
      }
    }
    if (gi < G) {   // the remaining (fewer than U) rows of this slice: one more batch, clamped and masked
      float4 v[U - 1];
#pragma unroll
      for (int u = 0; u < U - 1; ++u) {
        const int gr = gi + u * slices;
        v[u] = p4[(int64_t)(gr < G ? gr : G - 1) * nq + q];
      }
#pragma unroll
      for (int u = 0; u < U - 1; ++u) {
        const float m = gi + u * slices < G ? 1.0f : 0.0f;
        s[u & 3].x += m * v[u].x; s[u & 3].y += m * v[u].y; s[u & 3].z += m * v[u].z; s[u & 3].w += m * v[u].w;
      }
    }
    float* r = red + (slice * nq + q) * 4;
    r[0] = (s[0].x + s[1].x) + (s[2].x + s[3].x);
    r[1] = (s[0].y + s[1].y) + (s[2].y + s[3].y);
    r[2] = (s[0].z + s[1].z) + (s[2].z + s[3].z);
    r[3] = (s[0].w + s[1].w) + (s[2].w + s[3].w);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += THREADS) {
    float t = 0.0f;
    for (int sl = 0; sl < slices; ++sl) t += red[sl * 2 * D + c];
    tot[c] = t;
  }
  __syncthreads();
}

template <int THREADS, int NF>
__device__ __forceinline__ void reduce_partials_t(const float* part, int G, int D, float* red, float* tot) {
  PartialBatchT<NF> pb;
  partials_request_t<THREADS, NF>(part, G, D, pb);
  reduce_partials_finish_t<THREADS, NF>(part, G, D, pb, red, tot);
}

__device__ __forceinline__ void partials_request(const float* part, int G, int D, PartialBatch& pb) {
  partials_request_t<kRowThreads, kPartialFirst>(part, G, D, pb);
}
__device__ __forceinline__ void reduce_partials_finish(const float* part, int G, int D, const PartialBatch& pb,
                                                       float* red, float* tot) {
  reduce_partials_finish_t<kRowThreads, kPartialFirst>(part, G, D, pb, red, tot);
}
__device__ __forceinline__ void reduce_partials(const float* part, int G, int D, float* red, float* tot) {
  reduce_partials_t<kRowThreads, kPartialFirst>(part, G, D, red, tot);
}

// BatchNorm partial statistics are SHIFTED sums (round 3): a producer accumulates sum (y - K) and sum (y - K)^2 per
// column, K = the running mean of the BatchNorm that will consume them as the producer saw it (or 0), and records K in
// one extra row behind its G partial rows - layout [G + 1][2][D], row G = [K | unused].  mean = K + S1 / M and
// var = S2 / M - (S1 / M)^2 then cancel at the scale of |batch mean - K| (what the running mean tracks) instead of
// |batch mean|: E[y^2] - mean^2 in fp32 loses the variance of a column whose mean is 10^3 standard deviations away,
// nn.BatchNorm1d (what the reference calls, Welford) does not.  The consumer reads K from the partial buffer, never
// from the running mean itself: workgroup 0 of the consumer updates that while the others may still be starting.
__device__ __forceinline__ float partials_shift(const float* part, int G, int D, int c) {
  return part[(int64_t)G * 2 * D + c];
}
// tot: [2][D] totals of the shifted sums over M rows -> batch mean and biased variance of column c
__device__ __forceinline__ void bn_moments(const float* part, int G, int D, int M, const float* tot, int c, float& mean,
                                           float& var) {
  const float m1 = tot[c] / (float)M;
  mean = partials_shift(part, G, D, c) + m1;
  var = fmaxf(tot[D + c] / (float)M - m1 * m1, 0.0f);
}

// the same with the shift already in a register: the consumers request it (and gamma / beta) with their first loads - read
// after the reduction it is one more dependent round trip to the memory side in the middle of the prologue
__device__ __forceinline__ void bn_moments_k(float shift, int D, int M, const float* tot, int c, float& mean, float& var) {
  const float m1 = tot[c] / (float)M;
  mean = shift + m1;
  var = fmaxf(tot[D + c] / (float)M - m1 * m1, 0.0f);
}

__host__ __device__ inline int reduce_red_floats(int D, int threads = kRowThreads) {
  const int nq = 2 * D / 4;
  const int slices = threads / nq > 0 ? threads / nq : 1;
  return slices * 2 * D;  // red[slices][2D]
}
__host__ __device__ inline int reduce_scratch_floats(int D, int threads = kRowThreads) {
  return 2 * D + reduce_red_floats(D, threads);  // tot[2][D] + red
}

}  // namespace feta
