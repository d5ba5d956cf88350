// Probe: semantics of ds_read_b64_tr_b16 on gfx950 with PER-LANE addresses (round 6, fused TCN weight gradient).
// Model under test (cdna_hip_programming.md, LDS section): inside each 16-lane group, lane p supplies the address of four
// contiguous 16-bit elements; lane i receives element (i & 3) of the run supplied by lane 4 j + (i >> 2), j = 0 .. 3.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o tools/probe/tr_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void k_probe(const int* __restrict__ offs, uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint16_t img[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) img[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t addr = (uint32_t)(uintptr_t)(img) + (uint32_t)offs[threadIdx.x] * 2u;  // byte address in LDS
  typedef short v4s __attribute__((ext_vector_type(4)));
  v4s r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)r[j];
}

int main() {
  int h_off[64];
  srand(7);
  for (int l = 0; l < 64; ++l) h_off[l] = 4 * (rand() % 1000);  // element offsets, 8-byte aligned runs
  int* d_off;
  uint16_t* d_out;
  hipMalloc(&d_off, sizeof(h_off));
  hipMalloc(&d_out, 64 * 4 * 2);
  hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_off, d_out);
  uint16_t h_out[256];
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int g = l & ~15, i = l & 15;
      const int want = h_off[g + 4 * j + (i >> 2)] + (i & 3);
      if (h_out[l * 4 + j] != want) {
        if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, j, h_out[l * 4 + j], want);
        ++bad;
      }
    }
  printf("tr_probe: %s (%d mismatches of 256)\n", bad ? "MODEL WRONG" : "model confirmed", bad);
  return bad ? 1 : 0;
}
