// Development probe (not product): what bounds one time step of the matrix-pipe GRU forward recurrence?
// Variants of k_gru16m_fwd with parts removed, timed with HIP events at the decoder size (1,024 sequences: one
// wavefront per SIMD at most) and the encoder size (14,336).  Build: hipcc --offload-arch=gfx950 -O3 gru_probe.hip -o gru_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ACT(t, c, C, Sp, s) ((((int64_t)(t)) * (Sp) + (s)) * (C) + (c))
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

template <bool STORE, bool LOAD, bool ELEM, bool XSIDE, bool HSIDE, int PF>
__global__ void __launch_bounds__(64) k_probe(const float* __restrict__ X, const float* __restrict__ wih,
                                              const float* __restrict__ whh, const float* __restrict__ bias,
                                              float* __restrict__ O, int T, int64_t S, int64_t Sp) {
  const int lane = threadIdx.x & 63, j = lane & 15, b = lane >> 4;
  const int64_t s = (int64_t)blockIdx.x * 16 + j;
  const int dir = blockIdx.y;
  float aix[3][4], ahh[3][4];
  for (int g = 0; g < 3; ++g)
    for (int q = 0; q < 4; ++q) {
      aix[g][q] = wih[(g * 16 + j) * 16 + 4 * b + q];
      ahh[g][q] = whh[(g * 16 + j) * 16 + 4 * b + q];
    }
  f32x4 c_r, c_z, c_n, c_hn;
  for (int r = 0; r < 4; ++r) { c_r[r] = bias[4 * b + r]; c_z[r] = bias[16 + 4 * b + r]; c_n[r] = bias[32 + 4 * b + r]; c_hn[r] = bias[48 + 4 * b + r]; }
  float h[4] = {0.f, 0.f, 0.f, 0.f};
  float xs[PF][4];
  f32x4 g_r = c_r, g_z = c_z, g_n = c_n;
  auto load_x = [&](int slot, int step) {
    const int t = step < T ? step : 0;
    if (LOAD) {
      const float4 v = *reinterpret_cast<const float4*>(X + ACT(t, 4 * b, 16, Sp, s));
      xs[slot][0] = v.x; xs[slot][1] = v.y; xs[slot][2] = v.z; xs[slot][3] = v.w;
    } else {
      xs[slot][0] = xs[slot][1] = xs[slot][2] = xs[slot][3] = 0.01f * (float)(step & 3);
    }
  };
  auto input_half = [&](const float* xq) {
    g_r = c_r; g_z = c_z; g_n = c_n;
    if (XSIDE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        g_r = __builtin_amdgcn_mfma_f32_16x16x4f32(aix[0][q], xq[q], g_r, 0, 0, 0);
        g_z = __builtin_amdgcn_mfma_f32_16x16x4f32(aix[1][q], xq[q], g_z, 0, 0, 0);
        g_n = __builtin_amdgcn_mfma_f32_16x16x4f32(aix[2][q], xq[q], g_n, 0, 0, 0);
      }
    } else {
      for (int r = 0; r < 4; ++r) { g_r[r] += xq[r]; g_z[r] += xq[r]; g_n[r] += xq[r]; }
    }
  };
#pragma unroll
  for (int d = 0; d < PF; ++d) load_x(d, d);
  input_half(xs[0]);
  for (int step = 0; step < T; step += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const int st = step + d;
      f32x4 a_r = g_r, a_z = g_z, a_hn = c_hn;
      const f32x4 a_n = g_n;
      if (HSIDE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a_r = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[0][q], h[q], a_r, 0, 0, 0);
          a_z = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[1][q], h[q], a_z, 0, 0, 0);
          a_hn = __builtin_amdgcn_mfma_f32_16x16x4f32(ahh[2][q], h[q], a_hn, 0, 0, 0);
        }
      } else {
        for (int r = 0; r < 4; ++r) { a_r[r] += h[r]; a_z[r] += h[r]; a_hn[r] += h[r]; }
      }
      input_half(xs[(d + 1) % PF]);
      load_x(d, st + PF);
      const bool act = st < T;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float hnew;
        if (ELEM) {
          const float rr = sigm(a_r[r]), zz = sigm(a_z[r]);
          const float nn = tanh_(fmaf(rr, a_hn[r], a_n[r]));
          hnew = fmaf(zz, h[r] - nn, nn);
        } else {
          hnew = 0.25f * (a_r[r] + a_z[r]) + 0.125f * (a_hn[r] + a_n[r]);
        }
        h[r] = act ? hnew : h[r];
      }
      if (STORE && act) *reinterpret_cast<float4*>(O + ACT(st, dir * 16 + 4 * b, 32, Sp, s)) = make_float4(h[0], h[1], h[2], h[3]);
    }
  }
  if (!STORE) *reinterpret_cast<float4*>(O + ACT(0, dir * 16 + 4 * b, 32, Sp, s)) = make_float4(h[0], h[1], h[2], h[3]);
}

template <class K>
float time_it(K kern, dim3 grid, const float* X, const float* w1, const float* w2, const float* bs, float* O, int T, int64_t S, int64_t Sp) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, X, w1, w2, bs, O, T, S, Sp);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, X, w1, w2, bs, O, T, S, Sp);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.0f / reps;
}

int main() {
  const int T = 25;
  for (int64_t S : {1024, 14336}) {
    const int64_t Sp = S;
    float *X, *O, *w1, *w2, *bs;
    hipMalloc(&X, sizeof(float) * T * Sp * 16); hipMalloc(&O, sizeof(float) * T * Sp * 32);
    hipMalloc(&w1, 4 * 768); hipMalloc(&w2, 4 * 768); hipMalloc(&bs, 4 * 64);
    std::vector<float> hx((size_t)T * Sp * 16), hw(768), hb(64);
    for (auto& v : hx) v = 0.01f * (float)(rand() % 200 - 100);
    for (auto& v : hw) v = 0.002f * (float)(rand() % 200 - 100);
    for (auto& v : hb) v = 0.01f * (float)(rand() % 20 - 10);
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w1, hw.data(), 768 * 4, hipMemcpyHostToDevice); hipMemcpy(w2, hw.data(), 768 * 4, hipMemcpyHostToDevice);
    hipMemcpy(bs, hb.data(), 64 * 4, hipMemcpyHostToDevice);
    dim3 grid((unsigned)(S / 16), 2);
    printf("S=%ld  (us per launch, 25 steps)\n", (long)S);
#define RUN(name, ...) printf("  %-44s %8.2f\n", name, time_it(k_probe<__VA_ARGS__>, grid, X, w1, w2, bs, O, T, S, Sp));
    RUN("full (store,load,elem,x,h) PF4", true, true, true, true, true, 4)
    RUN("no store", false, true, true, true, true, 4)
    RUN("no load", true, false, true, true, true, 4)
    RUN("no store no load", false, false, true, true, true, 4)
    RUN("no elem (linear h)", true, true, false, true, true, 4)
    RUN("no x-side mfma", true, true, true, false, true, 4)
    RUN("no h-side mfma", true, true, true, true, false, 4)
    RUN("no mfma at all", true, true, true, false, false, 4)
    RUN("only mfma (no store/load/elem)", false, false, false, true, true, 4)
    RUN("only elem (no store/load/mfma)", false, false, true, false, false, 4)
    RUN("nothing (loop skeleton)", false, false, false, false, false, 4)
    RUN("full PF1", true, true, true, true, true, 1)
    hipFree(X); hipFree(O); hipFree(w1); hipFree(w2); hipFree(bs);
  }
  return 0;
}
