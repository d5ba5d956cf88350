// Development probe: does a hipGraph with two branches (captured with a fork / join over a second stream) run the branches
// concurrently, and what does one fork + join cost?  Main chain: NMAIN kernels of WG workgroups that spin ~US microseconds;
// side chain: NSIDE such kernels.  Prints us per replay for (a) everything in one chain, (b) the side chain forked.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/graph_fork_probe.hip -o tools/probe/graph_fork_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) spin(float* out, long long ticks) {
  const long long t0 = wall_clock64();
  float v = threadIdx.x;
  while (wall_clock64() - t0 < ticks) v = v * 1.0001f + 1.0f;
  out[blockIdx.x * 256 + threadIdx.x] = v;
}

static float replay(hipGraphExec_t ge, hipStream_t st, int n) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < n; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0f / n;
}

int main(int argc, char** argv) {
  const int nmain = argc > 1 ? atoi(argv[1]) : 10, nside = argc > 2 ? atoi(argv[2]) : 4, wg = argc > 3 ? atoi(argv[3]) : 112;
  const int us = argc > 4 ? atoi(argv[4]) : 12;
  const int wg_side = argc > 5 ? atoi(argv[5]) : wg;
  const long long ticks = (long long)us * 100;  // wall_clock64: 100 MHz
  float *a, *b; CK(hipMalloc(&a, 4096 * 256 * 4)); CK(hipMalloc(&b, 4096 * 256 * 4));
  hipStream_t st, s2; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  hipGraph_t g; hipGraphExec_t serial, forked;
  // (a) one chain
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nside; ++i) hipLaunchKernelGGL(spin, dim3(wg_side), dim3(256), 0, st, b, ticks);
  for (int i = 0; i < nmain; ++i) hipLaunchKernelGGL(spin, dim3(wg), dim3(256), 0, st, a, ticks);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&serial, g, nullptr, nullptr, 0));
  // (b) side chain on a second stream between a fork and a join
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(ef, st)); CK(hipStreamWaitEvent(s2, ef, 0));
  for (int i = 0; i < nside; ++i) hipLaunchKernelGGL(spin, dim3(wg_side), dim3(256), 0, s2, b, ticks);
  CK(hipEventRecord(ej, s2));
  for (int i = 0; i < nmain; ++i) hipLaunchKernelGGL(spin, dim3(wg), dim3(256), 0, st, a, ticks);
  CK(hipStreamWaitEvent(st, ej, 0));
  hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, st, a, 0);  // the join's consumer
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&forked, g, nullptr, nullptr, 0));
  // (c) main chain alone + the consumer (what a perfect overlap would cost)
  hipGraphExec_t alone;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nmain; ++i) hipLaunchKernelGGL(spin, dim3(wg), dim3(256), 0, st, a, ticks);
  hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, st, a, 0);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&alone, g, nullptr, nullptr, 0));
  const float ts = replay(serial, st, 200), tf = replay(forked, st, 200), ta = replay(alone, st, 200);
  printf("main %d + side %d kernels of %d / %d workgroups x ~%d us: one chain %.1f us, forked %.1f us, main chain alone %.1f us\n", nmain,
         nside, wg, wg_side, us, ts, tf, ta);
  return 0;
}
