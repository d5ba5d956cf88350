// Development probe: the time-resident TCN convolution kernels (k_tcn_conv_b) alone, at the C4 launch shape (T = 25,
// 114,688 sequences), timed with HIP events, plus cycle stamps of one tile of workgroup 0 (TCN_PROBE_STAMPS).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTCN_PROBE_STAMPS=6 -Iinclude -Ideepof_amd/csrc tools/probe/tcn_conv_probe.hip \
//        -o tools/probe/tcn_conv_probe     (includes k_tcn.hip itself: an instrumented copy of the kernels)
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../../deepof_amd/csrc/k_tcn.hip"

static char g_err[512];
void dof_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
int dof_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return DOF_ERR_LAUNCH; }
  return DOF_OK;
}
int dof_launch_sum_partials(const float*, int64_t, int, float*, int, hipStream_t) { return DOF_OK; }

static void stamps(const char* tag, bool wgrad) {
  unsigned long long h[256];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tcn_stamps), sizeof h);
  printf("%s: stamps of tile %d of workgroup 0 (cycles relative to loader wave 4's iteration start)\n", tag, TCN_PROBE_STAMPS);
  const unsigned long long t0 = h[128];
  for (int w = 0; w < 4; ++w) {
    printf("  compute wave %d:", w);
    for (int r = 0; r < 4; ++r) {
      printf("  r%d [", r);
      for (int j = 0; j < (wgrad ? 5 : 3); ++j) printf("%s%lld", j ? " " : "", (long long)(h[32 * w + 8 * r + j] - t0));
      printf("]");
    }
    printf("  end-barrier in %lld out %lld\n", (long long)(h[32 * w + 31] - t0), (long long)(h[32 * w + 30] - t0));
  }
  for (int w = 0; w < 4; ++w)
    printf("  loader wave %d: start %lld staged %lld issued %lld past-barrier %lld\n", w + 4, (long long)(h[128 + 16 * w] - t0),
           (long long)(h[128 + 16 * w + 1] - t0), (long long)(h[128 + 16 * w + 2] - t0), (long long)(h[128 + 16 * w + 3] - t0));
}

int main(int argc, char** argv) {
  const int T = 25, dil = argc > 1 ? atoi(argv[1]) : 2;
  printf("dilation %d\n", dil);
  const int64_t S = 8192 * 14, Sp = S;
  const size_t n = (size_t)T * Sp * 32;
  float *in, *y, *out, *out2, *src, *xprev, *w, *bias, *bnp, *coef, *partial, *wgp;
  hipMalloc(&in, n * 4); hipMalloc(&y, n * 4); hipMalloc(&out, n * 4); hipMalloc(&out2, n * 4); hipMalloc(&src, n * 4);
  hipMalloc(&xprev, n * 4);
  hipMalloc(&w, 32 * 32 * 4 * 4); hipMalloc(&bias, 128); hipMalloc(&bnp, 4 * 32 * 4); hipMalloc(&coef, 64 * 4);
  hipMalloc(&partial, 2048 * 96 * 4); hipMalloc(&wgp, (size_t)2 * 256 * DOF_OUTER_PARTIAL_FLOATS * 4);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(y, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(xprev, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<float> hw(32 * 32 * 4), hb(128, 0.0f);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.05f * ((float)((i * 40503u) & 0xff) / 128.0f - 1.0f);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  for (int c = 0; c < 32; ++c) { hb[c] = 0.0f; hb[32 + c] = 1.0f; hb[64 + c] = 1.0f; hb[96 + c] = 0.0f; }
  hipMemcpy(bnp, hb.data(), 128 * 4, hipMemcpyHostToDevice);
  hipMemset(coef, 0, 64 * 4); hipMemset(bias, 0, 128);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* tag, auto fn, double gbytes, bool wgrad) {
    for (int i = 0; i < 3; ++i) fn();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int it = 20;
    for (int i = 0; i < it; ++i) fn();
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us   %6.2f TB/s of %.2f GB algorithmic\n", tag, 1e3 * ms / it, gbytes / (ms / it) , gbytes);
    stamps(tag, wgrad);
  };
  const double GB = n * 4 / 1e9;
  timeit("fwd BN_IN (1R+1W)", [&] { dof_launch_tcn_conv(0, in, w, bias, bnp, nullptr, out, partial, 0, T, dil, S, Sp, 0, nullptr, nullptr, nullptr, nullptr, 1, 1); }, 2 * GB, false);
  timeit("fwd COMB (2R+2W)", [&] { dof_launch_tcn_conv_comb(in, y, bnp, out2, w, bias, out, partial, T, dil, S, Sp, 0, nullptr, 1, nullptr); }, 4 * GB, false);
  timeit("bwd BWD2+FUSE (3R+1W)", [&] { dof_launch_tcn_conv_bwd_bn(in, w, y, bnp, out, partial, nullptr, T, dil, S, Sp, 0, y, bnp, coef, 0); }, 4 * GB, false);
  timeit("bwd BWD2+FUSE+WGRAD (3R+1W)", [&] { dof_launch_tcn_conv_bwd_bn(in, w, y, bnp, out, partial, nullptr, T, dil, S, Sp, 0, y, bnp, coef, 0, wgp, 0, (int64_t)256 * DOF_OUTER_PARTIAL_FLOATS); }, 4 * GB, true);
  timeit("bwd TAIL+WGRAD (5R+2W)", [&] { dof_launch_tcn_conv_tail(in, w, y, bnp, coef, 0, src, nullptr, out2, src, nullptr, y, bnp, out, partial, nullptr, T, dil, S, Sp, 0, xprev, wgp, 0, (int64_t)256 * DOF_OUTER_PARTIAL_FLOATS); }, 7 * GB, true);
  return 0;
}
