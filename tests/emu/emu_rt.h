// TEST TOOL (not product): a minimal SIMT emulator so the *same* .hip kernel sources can be
// compiled with g++ and executed on the CPU inside pytest (-m "not gpu"), one fiber per HIP
// thread, with real __syncthreads / wave-shuffle / MFMA semantics.  There is no GPU in the
// build container; this is how kernel logic is debugged before it is sent to a real MI355X.
// It is never linked into libdeepof_hip.so and the deepof_amd package cannot load it.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__ __restrict
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

extern dim3 threadIdx, blockIdx, blockDim, gridDim;
extern unsigned char emu_dyn_smem[];

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }

void emu_sync_block();
float emu_shfl(float v, int src_lane, int width);
void emu_run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
typedef float emu_f32x4 __attribute__((vector_size(16)));
emu_f32x4 emu_mfma_16x16x4(float a, float b, emu_f32x4 c);
struct emu_bf16x8 { uint16_t v[8]; };
typedef float emu_f32x16 __attribute__((vector_size(64)));
emu_f32x16 emu_mfma_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c);
emu_f32x4 emu_mfma_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c);
void emu_lds_tr16(const uint16_t* p, uint16_t out[4]);  // ds_read_b64_tr_b16 (see dof_lds_tr16)

static inline void __syncthreads() { emu_sync_block(); }
static inline float __shfl_xor(float v, int mask, int width = 64) {
  int lane = (int)(threadIdx.x & 63);
  return emu_shfl(v, lane ^ mask, width);
}
static inline float __shfl(float v, int src, int width = 64) { return emu_shfl(v, src, width); }
static inline float __shfl_down(float v, unsigned d, int width = 64) {
  int lane = (int)(threadIdx.x & 63);
  int src = lane + (int)d;
  if ((src / width) != (lane / width)) src = lane;
  return emu_shfl(v, src, width);
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_16x16x4((a), (b), (c))

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

template <class K, class... A>
static inline void emu_launch(K kern, dim3 grid, dim3 block, A... args) {
  emu_run_grid(grid, block, [&]() { kern(args...); });
}
