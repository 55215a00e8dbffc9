// gfx950 device primitives used by every kernel in this directory: the f32 MFMA
// tile, wave shuffles and the single dynamic-LDS array.  Kernels are written
// against this header only (tools/simt swaps in a host emulation of the same API
// so tests can run the kernel source on a CPU).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// All LDS lives in ONE dynamic array (cdna_hip_programming.md: a second
// __shared__ object can serialise LDS-DMA waits); kernels carve it by hand.
extern __shared__ __attribute__((aligned(16))) float feta_lds[];

// Makes a (wave-uniform) pointer opaque to the optimizer at this point: loads through it are not hoisted above it.
// (the laundered value comes back as a generic pointer: FETA_GLOBAL restores the global address space, without which
// every access through it becomes a flat_load / flat_store and waits on both memory counters)
#define FETA_OPAQUE_PTR(p) asm volatile("" : "+s"(p))
// Makes a per-lane value opaque at this point.  Used on the LANE ID at the top of a graph / row-block loop: everything a
// lane derives from its id is invariant in such a loop, the compiler hoists all of it and holds hundreds of registers
// across the body (csrc/block_bwd.hip: up to 676 B of scratch per lane); laundered, the values are recomputed where used.
#define FETA_OPAQUE_LANE(x) asm volatile("" : "+v"(x))
typedef const float __attribute__((address_space(1)))* feta_gcf;   // keep accesses through a laundered pointer global_*

namespace feta {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain): lane l supplies
// A[l&15][l>>4] and B[l>>4][l&15]; register r of the result is D[4*(l>>4)+r][l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- bf16 storage (feta_bf16.h) ---------------------------------------------------------------------------
struct bf16_t {
  unsigned short bits;
};
struct __attribute__((aligned(8))) bf16x4_raw {
  bf16_t v[4];
};
typedef short bf16x4_mfma __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t x) { return __builtin_bit_cast(float, (unsigned)x.bits << 16); }
// round to nearest even; a plain cast (v_cvt_pk_bf16_f32 on gfx950) keeps a NaN a NaN, the integer trick does not
__device__ __forceinline__ bf16_t f2bf(float x) {
  bf16_t r;
  r.bits = __builtin_bit_cast(unsigned short, static_cast<__bf16>(x));
  return r;
}
// v_mfma_f32_16x16x16_bf16: D = A(16x16) B(16x16) + C, fp32 accumulate; lane l supplies A[l&15][4(l>>4) .. +3]
// and B[4(l>>4) .. +3][l&15]; register r of the result is D[4(l>>4)+r][l&15] (the layout of mfma16)
__device__ __forceinline__ f32x4 mfma16_bf16(const float (&a)[4], const float (&b)[4], f32x4 c) {
  bf16x4_mfma av, bv;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    av[i] = (short)f2bf(a[i]).bits;
    bv[i] = (short)f2bf(b[i]).bits;
  }
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, c, 0, 0, 0);
}

// Four k-consecutive bf16 operand values of one lane, already packed (two VGPRs): what the fused bf16 stack kernels
// keep in LDS / registers, so that an operand is rounded ONCE however many products it enters (feta_lp.h).
struct __attribute__((aligned(8))) bf16x4_pk {
  bf16x4_mfma v;
};
// two v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
__device__ __forceinline__ bf16x4_pk pack_bf16x4(float a, float b, float c, float d) {
  typedef __bf16 bf16x4_native __attribute__((ext_vector_type(4)));
  const f32x4 v = {a, b, c, d};
  bf16x4_pk r;
  r.v = __builtin_bit_cast(bf16x4_mfma, __builtin_convertvector(v, bf16x4_native));
  return r;
}
__device__ __forceinline__ bf16x4_pk pack_bf16x4_raw(bf16_t a, bf16_t b, bf16_t c, bf16_t d) {
  bf16x4_pk r;
  r.v[0] = (short)a.bits; r.v[1] = (short)b.bits; r.v[2] = (short)c.bits; r.v[3] = (short)d.bits;
  return r;
}
__device__ __forceinline__ float bf16x4_get(const bf16x4_pk& p, int i) {
  bf16_t t;
  t.bits = (unsigned short)p.v[i];
  return bf2f(t);
}
__device__ __forceinline__ f32x4 mfma16_bf16_pk(const bf16x4_pk& a, const bf16x4_pk& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.v, b.v, c, 0, 0, 0);
}

__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }

// Sum over the 16 lanes of a DPP row (lanes with equal lane >> 4); every lane of the row gets the
// total.  Four rotate-and-add steps on the VALU (row_ror:8,4,2,1) instead of four ds_bpermute
// round trips through the LDS crossbar per value.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// Sum over the 8 lanes of HALF a DPP row (lanes with equal lane >> 3): an xor butterfly on the VALU - quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror (lane i <-> 7 - i of the half row: after the two quad steps every lane of a quad
// holds the quad's sum).  Every lane of the group gets bit for bit the same total.  (A 64-element bf16 row is 8 lanes of
// 16 bytes: feta_ln.h.)
__device__ __forceinline__ float row8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
  return v;
}

// A wave's DS operations execute in order; this only stops the compiler from
// moving a wave-private LDS read above the write that another lane made.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS traffic only: global loads (and stores) in flight stay in flight.  __syncthreads()
// is a release / acquire on ALL address spaces, and on gfx9 one counter (vmcnt) covers loads and stores alike - once a
// global store has been issued, every later __syncthreads() also waits for every prefetch that is still travelling.
// Use it only where the data handed between the waves lives in LDS.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
// tanh(x) = 1 - 2 / (1 + e^{2x}): saturates correctly at +-1, absolute error ~1e-7
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));  // v_rcp_f32: 1 ulp
}

// 1-ulp hardware approximations (v_rcp_f32 / v_sqrt_f32 / v_rsq_f32), for values whose last bits do not
// reach the result (rotation angles of the Jacobi sweeps)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }

}  // namespace feta
