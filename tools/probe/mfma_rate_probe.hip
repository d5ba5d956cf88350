// Development probe: issue cost of v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x4_f32 on one wavefront per SIMD, with
// NACC independent accumulators issued round robin (NACC = 1: a dependent chain).  Prints ns and cycles (s_memtime) per MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_rate_probe.hip -o tools/probe/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool BF16>
__global__ void __launch_bounds__(256) k(float* out, int iters, long long* cyc) {
  f32x4 acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  union { unsigned w[4]; bf16x8 v; } ua, ub;
  for (int i = 0; i < 4; ++i) { ua.w[i] = 0x3f803f80u + threadIdx.x; ub.w[i] = 0x3f003f00u + threadIdx.x * 3; }
  float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        if constexpr (BF16) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[a], 0, 0, 0);
        else acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[a], 0, 0, 0);
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, bool BF16>
void run(const char* name, int blocks) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, BF16>), dim3(blocks), dim3(256), 0, 0, out, 10, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, BF16>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 8 * NACC;
  printf("  %-34s blocks %4d  %7.2f ns / MFMA / wave   %7.1f shader-clock ticks / MFMA\n", name, blocks, ms * 1e6 / n, (double)c / n);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {256, 512}) {   // 256-thread workgroups: 1 or 2 wavefronts per SIMD
    run<1, true>("bf16 16x16x32, 1 accumulator", blocks);
    run<2, true>("bf16 16x16x32, 2 accumulators", blocks);
    run<4, true>("bf16 16x16x32, 4 accumulators", blocks);
    run<8, true>("bf16 16x16x32, 8 accumulators", blocks);
    run<1, false>("f32 16x16x4, 1 accumulator", blocks);
    run<2, false>("f32 16x16x4, 2 accumulators", blocks);
    run<4, false>("f32 16x16x4, 4 accumulators", blocks);
  }
  return 0;
}
