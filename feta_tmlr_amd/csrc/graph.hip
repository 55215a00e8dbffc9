// Graph preprocessing: dense scaled Laplacian per graph from the batched edge list with
// the edge-list semantics of ChebConvDynamic.__norm__ (transformer/ChebNetDynamic.py:108-130,
// normalization='sym', lambda_max=2): self loops removed (:113), degree scattered on the
// source row (PyG get_laplacian), w = -deg_s^-1/2 deg_t^-1/2 (inf -> 0), the +1 Laplacian
// loops and the -1 loops of :125-127 cancel, duplicates are summed by the 'add' aggregation,
// and propagate() flows source -> target, so Lhat[b, t, s] += w  (out = Lhat @ x).
#include "feta_abi_common.h"
#include <feta_device.h>

namespace feta {

__global__ __launch_bounds__(256) void edge_degree_kernel(const int64_t* __restrict__ ei, int64_t E,
                                                           float* __restrict__ deg) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t s = ei[e], t = ei[E + e];
  if (s != t) atomicAdd(&deg[s], 1.0f);
}

__global__ __launch_bounds__(256) void edge_scatter_kernel(const int64_t* __restrict__ ei, int64_t E,
                                                            const int64_t* __restrict__ node_graph,
                                                            const int32_t* __restrict__ node_off,
                                                            const float* __restrict__ deg,
                                                            float* __restrict__ lhat, int N) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t s = ei[e], t = ei[E + e];
  if (s == t) return;
  const float ds = deg[s], dt = deg[t];
  const float w = (ds > 0.0f && dt > 0.0f) ? -rsqrtf(ds) * rsqrtf(dt) : 0.0f;
  const int64_t gidx = node_graph[s];
  const int off = node_off[gidx];
  const int ls = (int)(s - off), lt = (int)(t - off);
  if (ls < 0 || ls >= N || lt < 0 || lt >= N) return;  // edge leaving its graph: malformed batch
  atomicAdd(&lhat[(gidx * N + lt) * N + ls], w);
}

}  // namespace feta

using namespace feta;

extern "C" int feta_lhat_from_edges(const int64_t* edge_index, int64_t E, const int64_t* node_graph,
                                    const int32_t* node_off, float* deg, float* lhat, int B, int N,
                                    int64_t n_tot, feta_stream_t stream) {
  FETA_REQUIRE(node_graph && node_off && deg && lhat && B > 0 && N > 0 && n_tot > 0,
               "lhat_from_edges: bad arguments");
  if (E == 0) return FETA_OK;
  FETA_REQUIRE(edge_index != nullptr, "lhat_from_edges: null edge_index");
  const dim3 grid((unsigned)((E + 255) / 256)), block(256);
  auto k1 = edge_degree_kernel;
  hipLaunchKernelGGL(k1, grid, block, 0, (hipStream_t)stream, edge_index, E, deg);
  auto k2 = edge_scatter_kernel;
  hipLaunchKernelGGL(k2, grid, block, 0, (hipStream_t)stream, edge_index, E, node_graph, node_off,
                     deg, lhat, N);
  return check_launch("feta_lhat_from_edges");
}
