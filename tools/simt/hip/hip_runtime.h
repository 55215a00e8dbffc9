// Host SIMT emulation shim: stands in for <hip/hip_runtime.h> when the kernel
// sources under feta_tmlr_amd/csrc are compiled for the CPU (tools/simt/build.sh).
// TEST INFRASTRUCTURE ONLY - it lets tests/ run the real kernel source against the
// oracle without a GPU.  The product is built with hipcc and never sees this file.
//
// Model: one OS thread; every lane of a workgroup is a ucontext fiber; lanes run
// until they reach a wave-level collective (mfma / shuffle / wave LDS sync) or
// __syncthreads(), where they yield until the whole wave / workgroup has arrived.
// Collectives must therefore be wave-uniform, as on the hardware.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "simt-emu"; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__

namespace simt {
extern dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
struct WaveScratch {
  float a[64];
  float b[64];
  float c[64][4];
  float d[64][4];
};
WaveScratch& wave_scratch();   // scratch of the CURRENT lane's wave
void wave_barrier();           // all lanes of the current wave
void block_barrier();          // all lanes of the workgroup
void run_grid(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes);
constexpr int kLdsBytes = 160 * 1024;
}  // namespace simt

#define threadIdx simt::threadIdx_
#define blockIdx simt::blockIdx_
#define blockDim simt::blockDim_
#define gridDim simt::gridDim_

inline void __syncthreads() { simt::block_barrier(); }

inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline float __expf(float x) { return expf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }

struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

template <class K, class... A>
inline void simt_launch(K kernel, dim3 grid, dim3 block, size_t lds, A... args) {
  simt::run_grid([&]() { kernel(args...); }, grid, block, lds);
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  simt_launch(kernel, dim3(grid), dim3(block), (size_t)(lds), __VA_ARGS__)
