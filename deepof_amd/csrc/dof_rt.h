// Runtime glue shared by every kernel file of libdeepof_hip.
//
// Product build: hipcc --offload-arch=gfx950 (HIP runtime, real launches on a stream).
// DOF_EMU build: a g++ build of the SAME sources against tests/emu (a pytest-only SIMT emulator
// used to debug kernel logic in the GPU-less build container).  The emulator library is a test
// tool: it is not shipped, the deepof_amd package never loads it, and there is no CPU fallback.
#pragma once
#include <cstdint>
#include <type_traits>
#include <utility>

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

// v_mfma_f32_32x32x16_bf16 (gfx950): D (32 x 32 fp32) += A (32 x 16 bf16) B (16 x 32 bf16).  Operand of lane l: row (A) /
// column (B) l & 31, the eight consecutive k values 8 (l >> 5) .. + 7; D register v of lane l: row 8 (v / 4) + 4 (l >> 5) +
// v % 4, column l & 31.
#ifdef DOF_EMU
typedef emu_bf16x8 dof_bf16x8;
typedef emu_f32x16 dof_f32x16;
#define DOF_MFMA_32x32x16_BF16(a, b, c) emu_mfma_32x32x16_bf16((a), (b), (c))
#define DOF_MFMA_16x16x32_BF16(a, b, c) emu_mfma_16x16x32_bf16((a), (b), (c))
#else
typedef __bf16 dof_bf16x8 __attribute__((ext_vector_type(8)));
typedef float dof_f32x16 __attribute__((ext_vector_type(16)));
#define DOF_MFMA_32x32x16_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// v_mfma_f32_16x16x32_bf16 (gfx950): D (16 x 16 fp32) += A (16 x 32 bf16) B (32 x 16 bf16).  Operand of lane l: row (A) /
// column (B) l & 15, the eight consecutive k values 8 (l >> 4) .. + 7; D register r of lane l: row 4 (l >> 4) + r, column
// l & 15 (the layout of v_mfma_f32_16x16x4_f32).
#define DOF_MFMA_16x16x32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
// eight bf16 values (16 bytes, 8-byte aligned) from LDS as an MFMA operand
__device__ __forceinline__ dof_bf16x8 dof_ld_bf16x8(const uint16_t* p) {
  union { uint64_t h[2]; dof_bf16x8 v; } u;
  u.h[0] = reinterpret_cast<const uint64_t*>(p)[0];
  u.h[1] = reinterpret_cast<const uint64_t*>(p)[1];
  return u.v;
}
// the same from an address that is 4-byte aligned only
__device__ __forceinline__ dof_bf16x8 dof_ld_bf16x8_a4(const uint16_t* p) {
  union { uint32_t w[4]; dof_bf16x8 v; } u;
#pragma unroll
  for (int k = 0; k < 4; ++k) u.w[k] = reinterpret_cast<const uint32_t*>(p)[k];
  return u.v;
}
// eight bf16 values from a 16-byte aligned LDS address: one ds_read_b128
__device__ __forceinline__ dof_bf16x8 dof_ld_bf16x8_16(const uint16_t* p) { return *reinterpret_cast<const dof_bf16x8*>(p); }
// ds_read_b64_tr_b16 (gfx950): the transposing LDS read.  Inside each 16-lane group, lane q supplies the (8-byte aligned)
// address of four contiguous 16-bit elements; lane i receives element i & 3 of the run supplied by lane 4 j + (i >> 2) as
// its element j (j = 0 .. 3) -- the four runs of a row j make a 16-element row, lane i reads column i of the 4 x 16 block
// (tools/probe/tr_probe.hip checks this model with per-lane addresses on the device).  w0 = elements 0, 1; w1 = 2, 3.
__device__ __forceinline__ void dof_lds_tr16(const uint16_t* p, uint32_t& w0, uint32_t& w1) {
#ifdef DOF_EMU
  uint16_t o[4];
  emu_lds_tr16(p, o);
  w0 = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
  w1 = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
#else
  typedef short dof_v4s __attribute__((ext_vector_type(4)));
  union { dof_v4s v; uint32_t w[2]; } u;
  u.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dof_v4s*)(p));
  w0 = u.w[0];
  w1 = u.w[1];
#endif
}
// fp32 = hi + mid + lo EXACTLY with three bf16 pieces: each piece is the top 16 bits of what is left (truncation keeps the
// remainder exactly representable: 8 + 8 + 8 significand bits).  dof_bf16_rest: the value with its top piece removed;
// dof_pack_hi16: the top pieces of two values as one word (first argument in the low half) -- one v_perm_b32.
__device__ __forceinline__ float dof_bf16_rest(float v) {
  return v - __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & 0xFFFF0000u);
}
__device__ __forceinline__ uint32_t dof_pack_hi16(float lo_half, float hi_half) {
#ifdef DOF_EMU
  return (__builtin_bit_cast(uint32_t, lo_half) >> 16) | (__builtin_bit_cast(uint32_t, hi_half) & 0xFFFF0000u);
#else
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi_half), __builtin_bit_cast(uint32_t, lo_half), 0x07060302u);
#endif
}

// an MFMA operand of eight bf16 values from four packed words (word w = elements 2w (low half) and 2w + 1)
__device__ __forceinline__ dof_bf16x8 dof_mk_bf16x8(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  union { uint32_t w[4]; dof_bf16x8 v; } u;
  u.w[0] = w0; u.w[1] = w1; u.w[2] = w2; u.w[3] = w3;
  return u.v;
}
// The three bf16 pieces of four fp32 values as packed words: out[p][0] = pieces p of v[0], v[1]; out[p][1] = of v[2], v[3].
// 4 VALU operations per value + 6 packs: v = piece0 + piece1 + piece2 exactly (see dof_bf16_rest).
__device__ __forceinline__ void dof_split3x4(const float (&v)[4], uint32_t (&out)[3][2]) {
  float r1[4], r2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { r1[i] = dof_bf16_rest(v[i]); r2[i] = dof_bf16_rest(r1[i]); }
  out[0][0] = dof_pack_hi16(v[0], v[1]);   out[0][1] = dof_pack_hi16(v[2], v[3]);
  out[1][0] = dof_pack_hi16(r1[0], r1[1]); out[1][1] = dof_pack_hi16(r1[2], r1[3]);
  out[2][0] = dof_pack_hi16(r2[0], r2[1]); out[2][1] = dof_pack_hi16(r2[2], r2[3]);
}

// 1: the HIP runtime is underneath (RCCL can be opened); 0: the host-only emulation build
#ifdef DOF_EMU
#define DOF_HAS_DEVICE_RUNTIME 0
#else
#define DOF_HAS_DEVICE_RUNTIME 1
#endif

// 16-byte streaming (non-temporal) stores of four words, and the round-to-nearest products / sums the compiler must
// not contract into FMAs (values compared bit for bit with numpy / torch op-by-op arithmetic).  The kernel sources use
// these names only: everything that differs between the device build and the pytest-only emulation build lives in
// this header.
#ifdef DOF_EMU
__device__ __forceinline__ void dof_st_stream4(float* out, const float (&v)[4]) {
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}
__device__ __forceinline__ void dof_st_stream4(uint32_t* out, const uint32_t (&w)[4]) {
  out[0] = w[0]; out[1] = w[1]; out[2] = w[2]; out[3] = w[3];
}
__device__ __forceinline__ float dof_fmul_rn(float a, float b) { volatile float r = a * b; return r; }
__device__ __forceinline__ float dof_fadd_rn(float a, float b) { volatile float r = a + b; return r; }
__device__ __forceinline__ double dof_dmul_rn(double a, double b) { volatile double r = a * b; return r; }
__device__ __forceinline__ double dof_dadd_rn(double a, double b) { volatile double r = a + b; return r; }
#else
__device__ __forceinline__ void dof_st_stream4(float* out, const float (&v)[4]) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 pack = {v[0], v[1], v[2], v[3]};
  __builtin_nontemporal_store(pack, reinterpret_cast<f32x4*>(out));
}
__device__ __forceinline__ void dof_st_stream4(uint32_t* out, const uint32_t (&w)[4]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 pack = {w[0], w[1], w[2], w[3]};
  __builtin_nontemporal_store(pack, reinterpret_cast<u32x4*>(out));
}
__device__ __forceinline__ float dof_fmul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float dof_fadd_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double dof_dmul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double dof_dadd_rn(double a, double b) { return __dadd_rn(a, b); }
#endif

// a lambda that must be inlined at every call site (a step body called from the unrolled loop AND from the remainder
// group is otherwise emitted as a function: its by-reference captures then live in scratch memory)
#define DOF_INLINE_LAMBDA __attribute__((always_inline))
#ifdef DOF_EMU
#define DOF_SCHED_FENCE() ((void)0)
#else
#define DOF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// Orders the LDS accesses of ONE wavefront that owns a region of LDS by itself (write the tile, read it transposed): the
// hardware executes a wavefront's DS instructions in issue order, so all it takes is that the compiler keeps them in
// program order (a wavefront-scope fence emits no instruction).  The emulator runs lanes as fibres and needs a real
// rendezvous of the wavefront's lanes (a shuffle is one).
#ifdef DOF_EMU
#define DOF_WAVE_LDS_ORDER() ((void)emu_shfl(0.0f, 0, 64))
#else
#define DOF_WAVE_LDS_ORDER() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#endif
// compiler-only memory fence: loads after it are not merged with / hoisted above earlier ones
#define DOF_MEM_FENCE() asm volatile("" ::: "memory")

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

// compile-time unrolled loop: dof_static_for<N>([&](auto kc) { constexpr int k = decltype(kc)::value; ... });
template <class F, int... I>
__device__ __forceinline__ void dof_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void dof_static_for(F&& f) {
  dof_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Value held by lane K of this lane's GROUP-lane group (GROUP = 16 or 8, groups aligned to DPP
// rows).  gfx950: v_mov_b32_dpp row_newbcast:K -- a full-rate VALU op, no LDS crossbar traffic.
template <int K, int GROUP>
__device__ __forceinline__ float dof_gbcast(float v) {
#ifdef DOF_EMU
  const int lane = (int)(threadIdx.x & 63);
  return __shfl(v, (lane & ~(GROUP - 1)) | K);
#else
  static_assert(GROUP == 16 || GROUP == 8, "group must be 8 or 16 lanes");
  const int iv = __builtin_bit_cast(int, v);
  if constexpr (GROUP == 16) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(iv, 0x150 + K, 0xf, 0xf, true));
  } else {
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(iv, 0x150 + K, 0xf, 0xf, true));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(iv, 0x150 + K + 8, 0xf, 0xf, true));
    return (threadIdx.x & 8) ? hi : lo;
  }
#endif
}

// Lane permutations inside a 16-lane DPP row (full-rate VALU modifiers, no LDS crossbar): CTRL = 0xB1 swaps
// neighbours (lane ^ 1), 0x4E swaps pairs (lane ^ 2), 0x141 mirrors each half row (i <-> 7 - i), 0x140 mirrors the
// row (i <-> 15 - i).  Applied in that order, "v op= perm(v)" leaves every lane with the reduction over its row, and
// all 16 lanes hold bitwise the same value (each step combines the same two operands in both partners).
template <int CTRL>
__device__ __forceinline__ float dof_dpp_perm(float v) {
#ifdef DOF_EMU
  const int lane = (int)(threadIdx.x & 63);
  const int src = CTRL == 0xB1 ? (lane ^ 1) : CTRL == 0x4E ? (lane ^ 2)
                : CTRL == 0x141 ? ((lane & ~7) | (7 - (lane & 7))) : ((lane & ~15) | (15 - (lane & 15)));
  return __shfl(v, src);
#else
  static_assert(CTRL == 0xB1 || CTRL == 0x4E || CTRL == 0x141 || CTRL == 0x140, "row permutation");
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
#endif
}
__device__ __forceinline__ float dof_row16_sum(float v) {
  v += dof_dpp_perm<0xB1>(v);
  v += dof_dpp_perm<0x4E>(v);
  v += dof_dpp_perm<0x141>(v);
  v += dof_dpp_perm<0x140>(v);
  return v;
}
__device__ __forceinline__ float dof_row16_max(float v) {
  v = fmaxf(v, dof_dpp_perm<0xB1>(v));
  v = fmaxf(v, dof_dpp_perm<0x4E>(v));
  v = fmaxf(v, dof_dpp_perm<0x141>(v));
  v = fmaxf(v, dof_dpp_perm<0x140>(v));
  return v;
}

// One sequence's C channels at one time step are 4C contiguous bytes, aligned to 16 (C % 4 == 0), 8 (C even) or 4: move
// them as the widest words that alignment allows (scalar dword accesses at a 4C-byte lane stride touch 64 cache lines
// per instruction).
template <int C>
__device__ __forceinline__ void dof_ld_row(const float* __restrict__ p, float* v) {
  if constexpr (C % 4 == 0) {
#pragma unroll
    for (int i = 0; i < C / 4; ++i) {
      const float4 q = reinterpret_cast<const float4*>(p)[i];
      v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
  } else if constexpr (C % 2 == 0) {
#pragma unroll
    for (int i = 0; i < C / 2; ++i) {
      const float2 q = reinterpret_cast<const float2*>(p)[i];
      v[2 * i] = q.x; v[2 * i + 1] = q.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < C; ++i) v[i] = p[i];
  }
}
template <int C>
__device__ __forceinline__ void dof_st_row(float* __restrict__ p, const float* v) {
  if constexpr (C % 4 == 0) {
#pragma unroll
    for (int i = 0; i < C / 4; ++i)
      reinterpret_cast<float4*>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else if constexpr (C % 2 == 0) {
#pragma unroll
    for (int i = 0; i < C / 2; ++i) reinterpret_cast<float2*>(p)[i] = make_float2(v[2 * i], v[2 * i + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < C; ++i) p[i] = v[i];
  }
}

// two floats as one 8-byte store (8-byte aligned address)
__device__ __forceinline__ void dof_st_pair(float* __restrict__ p, float a, float b) {
  *reinterpret_cast<uint64_t*>(p) = (uint64_t)__builtin_bit_cast(uint32_t, a) | ((uint64_t)__builtin_bit_cast(uint32_t, b) << 32);
}

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
  __shared__ float part[32][9];
  const int tid = threadIdx.x;
  const int col = tid & 31, seg = tid >> 5;  // 8 segments of 32 rows per column
#pragma unroll
  for (int c0 = 0; c0 < NV; c0 += CH) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < CH; ++v)
      if (c0 + v < NV) tile[v][tid] = vals[c0 + v];
    __syncthreads();
    if (col < CH) {
      float acc = 0.0f;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) acc += tile[col][seg * 32 + i];
      part[col][seg] = acc;
    }
    __syncthreads();
    if (tid < CH && c0 + tid < NV) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += part[tid][k];
      out[c0 + tid] = acc;
    }
  }
}
