// Runtime glue shared by every kernel file of libdeepof_hip.
//
// Product build: hipcc --offload-arch=gfx950 (HIP runtime, real launches on a stream).
// DOF_EMU build: a g++ build of the SAME sources against tests/emu (a pytest-only SIMT emulator
// used to debug kernel logic in the GPU-less build container).  The emulator library is a test
// tool: it is not shipped, the deepof_amd package never loads it, and there is no CPU fallback.
#pragma once
#include <cstdint>

#ifdef DOF_EMU
#include "emu_rt.h"
#define DOF_LAUNCH(kernel, grid, block, stream, ...) emu_launch(kernel, dim3 grid, dim3 block, __VA_ARGS__)
typedef emu_f32x4 dof_f32x4;
#else
#include <hip/hip_runtime.h>
#define DOF_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3 grid, dim3 block, 0, stream, __VA_ARGS__)
typedef float dof_f32x4 __attribute__((ext_vector_type(4)));
#endif

#ifdef DOF_EMU
#define DOF_SCHED_FENCE() ((void)0)
#else
#define DOF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// Wave-uniform read-only weights: viewing them through the constant address space makes every
// uniform load a scalar-cache s_load (no VMEM traffic, weights arrive in SGPRs and feed v_fmac
// directly).  dof_opaque_zero() is an SGPR the optimiser cannot see through; adding it to the
// base inside a time loop keeps the loads inside the loop instead of being hoisted and spilled.
#ifdef DOF_EMU
typedef const float* dof_cfp;
static inline dof_cfp dof_cw(const float* p) { return p; }
static inline int dof_opaque_zero() { return 0; }
#else
typedef const float __attribute__((address_space(4)))* dof_cfp;
__device__ __forceinline__ dof_cfp dof_cw(const float* p) { return (dof_cfp)(uintptr_t)p; }
__device__ __forceinline__ int dof_opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}
#endif

// Index of element (time t, sequence s, channel c) of a time-major activation: channel-minor
// [t][s][c] ("AoS").  A thread that owns a sequence reads/writes its C channels as one contiguous
// 4C-byte run (vectorised), and a 16-lane group that owns the 16 hidden units of one sequence is
// perfectly coalesced; Sp = sequence count padded to 64.
#define ACT(t, c, C, Sp, s) ((((int64_t)(t)) * (Sp) + (s)) * (C) + (c))

#define DOF_OK 0
#define DOF_ERR_ARG (-1)
#define DOF_ERR_UNSUPPORTED (-2)
#define DOF_ERR_LAUNCH (-3)
#define DOF_ERR_STATE (-4)

void dof_set_error(const char* fmt, ...);
int dof_check_launch(const char* what);

static inline int64_t dof_pad64(int64_t n) { return (n + 63) / 64 * 64; }
static inline unsigned dof_cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dof_rcp(float x) {
#ifdef DOF_EMU
  return 1.0f / x;
#else
  return __builtin_amdgcn_rcpf(x);  // v_rcp_f32, 1 ulp
#endif
}
__device__ __forceinline__ float dof_sigmoid(float x) { return dof_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float dof_tanh(float x) {
  // 1 - 2/(1+e^{2x}); saturates cleanly at +-1 for large |x|
  return 1.0f - 2.0f * dof_rcp(1.0f + __expf(2.0f * x));
}
__device__ __forceinline__ float dof_softplus(float x) {
  // torch.nn.functional.softplus, beta=1, threshold=20
  return x > 20.0f ? x : log1pf(__expf(x));
}

// Block-wide column sums of per-thread vectors.  Every thread of a 256-thread block contributes
// vals[0..NV); thread v < NV leaves with the sum over the block of vals[v] and writes it to
// out[v] (per-block partials; a fixed-order finalize pass makes the result run-to-run stable).
template <int NV>
__device__ __forceinline__ void dof_block_colsum(const float* vals, float* out) {
  constexpr int CH = NV < 32 ? NV : 32;
  __shared__ float tile[32][257];
  const int tid = threadIdx.x;
#pragma unroll
  for (int c0 = 0; c0 < NV; c0 += CH) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < CH; ++v)
      if (c0 + v < NV) tile[v][tid] = vals[c0 + v];
    __syncthreads();
    if (tid < CH && c0 + tid < NV) {
      float acc = 0.0f;
      for (int i = 0; i < 256; ++i) acc += tile[tid][i];
      out[c0 + tid] = acc;
    }
  }
}
