// Development probe (not product): the matrix-pipe GRU kernels of the C2 step (deepof_amd/csrc/k_grum16.inc.h -- the
// product kernel text itself) timed in isolation at the C2 launch geometry (two streams of 14,336 sequences, 25 steps,
// both directions) beside round 4's fp32-MFMA kernels (gru_fp32_mfma.inc.h), outputs compared; the (32 -> 8) backward
// kernel is checked against a double-precision host evaluation of one tile.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deepof_amd/csrc -I include -I tools/probe tools/probe/gru16_probe.hip -o tools/probe/gru16_probe
// Run:   tools/probe/gru16_probe [sequences per stream]      (results of round 5: profiles/r05_gru_probe.txt)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dof_rt.h"
void dof_set_error(const char*, ...) {}
int dof_check_launch(const char*) { return 0; }
namespace {
#include "k_grum16.inc.h"
#include "gru_fp32_mfma.inc.h"   // round 4's fp32-MFMA kernels, for the comparison
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Bufs {
  float *X[2], *O[2], *GS8[2], *dO[2], *dX[2], *wg[2], *O8[2];
  int* len[2];
  float *w16, *w8;
  int64_t S, Sp;
  int T;
};

static float* dalloc(size_t n) { float* p; CK(hipMalloc(&p, n * sizeof(float))); CK(hipMemset(p, 0, n * sizeof(float))); return p; }
static void fill(float* d, size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  srand(seed);
  for (auto& v : h) v = scale * (float)(rand() % 2001 - 1000) * 1e-3f;
  CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

template <class F>
static float time_us(F launch, int reps = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0f / reps;
}

static double max_diff(const float* d_a, const std::vector<float>& ref, size_t n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), d_a, n * 4, hipMemcpyDeviceToHost));
  double m = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double d = fabs((double)h[i] - (double)ref[i]);
    if (!(d <= m)) m = d;   // (NaN propagates)
  }
  return m;
}
static std::vector<float> fetch(const float* d, size_t n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  return h;
}

int main(int argc, char** argv) {
  const int T = 25;
  const int64_t S = argc > 1 ? atol(argv[1]) : 14336, Sp = dof_pad64(S);
  Bufs B;
  B.S = S; B.Sp = Sp; B.T = T;
  // weights: [wih(48x16) whh(48x16) bih(48) bhh(48)] x 2 directions for the 16 -> 16 layer; 32 -> 8 layer: wih(24x32) whh(24x8) bih(24) bhh(24)
  const size_t W16 = 768 + 768 + 48 + 48, W8 = 768 + 192 + 24 + 24;
  B.w16 = dalloc(2 * W16); B.w8 = dalloc(2 * W8);
  fill(B.w16, 2 * W16, 0.25f, 1); fill(B.w8, 2 * W8, 0.25f, 2);
  for (int k = 0; k < 2; ++k) {
    B.X[k] = dalloc((size_t)T * Sp * 16); fill(B.X[k], (size_t)T * Sp * 16, 1.0f, 10 + k);
    B.O[k] = dalloc((size_t)T * Sp * 32);
    B.O8[k] = dalloc((size_t)T * Sp * 16);
    B.GS8[k] = dalloc((size_t)2 * T * 32 * Sp);
    B.dO[k] = dalloc((size_t)T * Sp * 32); fill(B.dO[k], (size_t)T * Sp * 32, 0.1f, 20 + k);
    B.dX[k] = dalloc((size_t)2 * T * 16 * Sp);
    B.wg[k] = dalloc((size_t)2 * ((S + 15) / 16) * GRU16_WG_FLOATS);
    CK(hipMalloc(&B.len[k], Sp * sizeof(int)));
    std::vector<int> hl(Sp, T);
    CK(hipMemcpy(B.len[k], hl.data(), Sp * sizeof(int), hipMemcpyHostToDevice));
  }
  auto stream16 = [&](int k) {
    Gru16mStream a{};
    a.X = B.X[k]; a.len = B.len[k];
    a.wih0 = B.w16; a.whh0 = B.w16 + 768; a.bih0 = B.w16 + 1536; a.bhh0 = B.w16 + 1584;
    a.wih1 = B.w16 + W16; a.whh1 = B.w16 + W16 + 768; a.bih1 = B.w16 + W16 + 1536; a.bhh1 = B.w16 + W16 + 1584;
    a.O = B.O[k]; a.GS = nullptr; a.dO = B.dO[k]; a.dX = B.dX[k]; a.wg_partial = B.wg[k];
    a.S = S; a.Sp = Sp;
    return a;
  };
  auto stream8 = [&](int k) {   // input = the 16 -> 16 layer's output (32 channels)
    Gru16mStream a{};
    a.X = B.O[k]; a.len = B.len[k];
    a.wih0 = B.w8; a.whh0 = B.w8 + 768; a.bih0 = B.w8 + 960; a.bhh0 = B.w8 + 984;
    a.wih1 = B.w8 + W8; a.whh1 = B.w8 + W8 + 768; a.bih1 = B.w8 + W8 + 960; a.bhh1 = B.w8 + W8 + 984;
    a.O = B.O8[k]; a.GS = B.GS8[k];
    a.S = S; a.Sp = Sp;
    return a;
  };
  const Gru16mStream a16 = stream16(0), b16 = stream16(1), a8 = stream8(0), b8 = stream8(1);
  const size_t nO = (size_t)T * Sp * 32, ndX = (size_t)2 * T * 16 * Sp, nwg = (size_t)2 * ((S + 15) / 16) * GRU16_WG_FLOATS;
  printf("S = %ld per stream, 2 streams, T = %d (us per launch)\n", (long)S, T);

  // ---- forward 16 -> 16
  std::vector<float> refO;
#define FWD(NT, WPE)                                                                                                  \
  {                                                                                                                   \
    CK(hipMemset(B.O[0], 0, nO * 4));                                                                                 \
    const float us = time_us([&] { hipLaunchKernelGGL((k_gru16m_fwd<NT, WPE>), dim3(dof_cdiv(S, 16 * NT), 2, 2), dim3(64), 0, 0, a16, b16, T); }); \
    if (refO.empty()) refO = fetch(B.O[0], nO);                                                                       \
    printf("  k_gru16m_fwd<NT=%d, WPE=%d>  %8.2f us   max |O - O_first| = %.3g\n", NT, WPE, us, max_diff(B.O[0], refO, nO)); \
  }
  FWD(1, 1) FWD(1, 4) FWD(2, 2)   // round 4's kernel: one / two tiles per wavefront, register bounds
#define FWDX(WPE)                                                                                                     \
  {                                                                                                                   \
    CK(hipMemset(B.O[0], 0, nO * 4));                                                                                 \
    const float us = time_us([&] { hipLaunchKernelGGL((k_gru16x_fwd<WPE, false>), dim3(dof_cdiv(S, 16), 2, 2), dim3(64), 0, 0, a16, b16, T); }); \
    printf("  k_gru16x_fwd<WPE=%d> (bf16x3)  %8.2f us   max |O - O_first| = %.3g\n", WPE, us, max_diff(B.O[0], refO, nO)); \
  }
  FWDX(3) FWDX(4)

  // ---- forward 32 -> 8 (reads the layer above's output)
  {
    const float us = time_us([&] { hipLaunchKernelGGL(k_gru8m_fwd, dim3(dof_cdiv(S, 16), 2, 2), dim3(64), 0, 0, a8, b8, T); });
    printf("  k_gru8m_fwd                  %8.2f us\n", us);
  }

  {
    std::vector<float> ref8 = fetch(B.O8[0], (size_t)T * Sp * 16);
    CK(hipMemset(B.O8[0], 0, (size_t)T * Sp * 16 * 4));
    const float us = time_us([&] { hipLaunchKernelGGL((k_gru8x_fwd<4>), dim3(dof_cdiv(S, 16), 2, 2), dim3(64), 0, 0, a8, b8, T); });
    printf("  k_gru8x_fwd<4> (bf16x3)      %8.2f us   max |O - O_m| = %.3g\n", us, max_diff(B.O8[0], ref8, (size_t)T * Sp * 16));
    const float us3 = time_us([&] { hipLaunchKernelGGL((k_gru8x_fwd<3>), dim3(dof_cdiv(S, 16), 2, 2), dim3(64), 0, 0, a8, b8, T); });
    printf("  k_gru8x_fwd<3> (bf16x3)      %8.2f us\n", us3);
  }
  // ---- backward 32 -> 8 with recompute, checked against a double-precision host evaluation of tile 0 (16 sequences)
  {
    float* dHfin = dalloc((size_t)16 * Sp); fill(dHfin, (size_t)16 * Sp, 0.5f, 77);
    float* dX8[2]; float* wg8[2];
    const size_t ndX8 = (size_t)2 * T * 32 * Sp, nwg8 = (size_t)2 * ((S + 15) / 16) * GRU8X_WG_FLOATS;
    for (int k = 0; k < 2; ++k) { dX8[k] = dalloc(ndX8); wg8[k] = dalloc(nwg8); }
    Gru16mStream c8 = a8, d8 = b8;
    c8.dO = dHfin; c8.dX = dX8[0]; c8.wg_partial = wg8[0];
    d8.dO = dHfin; d8.dX = dX8[1]; d8.wg_partial = wg8[1];
    hipLaunchKernelGGL((k_gru8x_fwd<4>), dim3(dof_cdiv(S, 16), 2, 2), dim3(64), 0, 0, a8, b8, T);   // O8 of the x kernel
    const float us = time_us([&] { hipLaunchKernelGGL(k_gru8x_bwd, dim3(dof_cdiv(S, 16), 2, 2), dim3(64), 0, 0, c8, d8, T); });
    // host reference
    std::vector<float> hX = fetch(B.O[0], nO), hO = fetch(B.O8[0], (size_t)T * Sp * 16), hd = fetch(dHfin, (size_t)16 * Sp);
    std::vector<float> hw = fetch(B.w8, 2 * W8), gdx = fetch(dX8[0], ndX8), gwg = fetch(wg8[0], nwg8);
    double worst_dx = 0.0, worst_wg = 0.0, scale_dx = 0.0, scale_wg = 0.0;
    for (int dir = 0; dir < 2; ++dir) {
      const float* wih = hw.data() + dir * W8; const float* whh = wih + 768; const float* bih = wih + 960; const float* bhh = wih + 984;
      std::vector<double> part(GRU8X_WG_FLOATS, 0.0);
      for (int sq = 0; sq < 16; ++sq) {
        double dh[8];
        for (int u = 0; u < 8; ++u) dh[u] = hd[(size_t)(dir * 8 + u) * Sp + sq];
        for (int step = T - 1; step >= 0; --step) {
          const int t = dir ? (T - 1 - step) : step, tp = dir ? t + 1 : t - 1;
          double x[32], hp[8];
          for (int c = 0; c < 32; ++c) x[c] = hX[((size_t)t * Sp + sq) * 32 + c];
          for (int u = 0; u < 8; ++u) hp[u] = step > 0 ? hO[((size_t)tp * Sp + sq) * 16 + dir * 8 + u] : 0.0;
          double gr[8], gz[8], gn[8], gh[8], dhn[8];
          for (int u = 0; u < 8; ++u) {
            double ar = bih[u] + bhh[u], az = bih[8 + u] + bhh[8 + u], an = bih[16 + u], ah = bhh[16 + u];
            for (int c = 0; c < 32; ++c) { ar += wih[u * 32 + c] * x[c]; az += wih[(8 + u) * 32 + c] * x[c]; an += wih[(16 + u) * 32 + c] * x[c]; }
            for (int v = 0; v < 8; ++v) { ar += whh[u * 8 + v] * hp[v]; az += whh[(8 + u) * 8 + v] * hp[v]; ah += whh[(16 + u) * 8 + v] * hp[v]; }
            const double r = 1.0 / (1.0 + exp(-ar)), z = 1.0 / (1.0 + exp(-az)), nn = tanh(an + r * ah);
            const double dht = dh[u], dn = dht * (1.0 - z), dz = dht * (hp[u] - nn), dnp = dn * (1.0 - nn * nn);
            gr[u] = dnp * ah * r * (1.0 - r); gz[u] = dz * z * (1.0 - z); gn[u] = dnp; gh[u] = dnp * r; dhn[u] = dht * z;
          }
          for (int v = 0; v < 8; ++v) for (int u = 0; u < 8; ++u) dhn[v] += whh[u * 8 + v] * gr[u] + whh[(8 + u) * 8 + v] * gz[u] + whh[(16 + u) * 8 + v] * gh[u];
          for (int c = 0; c < 32; ++c) {
            double dx = 0.0;
            for (int u = 0; u < 8; ++u) dx += wih[u * 32 + c] * gr[u] + wih[(8 + u) * 32 + c] * gz[u] + wih[(16 + u) * 32 + c] * gn[u];
            const double got = gdx[(size_t)dir * T * 32 * Sp + ((size_t)t * Sp + sq) * 32 + c];
            worst_dx = fmax(worst_dx, fabs(got - dx)); scale_dx = fmax(scale_dx, fabs(dx));
          }
          for (int u = 0; u < 8; ++u) {
            for (int c = 0; c < 32; ++c) { part[u * 32 + c] += gr[u] * x[c]; part[(8 + u) * 32 + c] += gz[u] * x[c]; part[(16 + u) * 32 + c] += gn[u] * x[c]; }
            for (int v = 0; v < 8; ++v) { part[768 + u * 8 + v] += gr[u] * hp[v]; part[768 + (8 + u) * 8 + v] += gz[u] * hp[v]; part[768 + (16 + u) * 8 + v] += gh[u] * hp[v]; }
            part[960 + u] += gr[u]; part[968 + u] += gz[u]; part[976 + u] += gn[u]; part[984 + u] += gh[u];
          }
          for (int u = 0; u < 8; ++u) dh[u] = dhn[u];
        }
      }
      const float* got = gwg.data() + (size_t)dir * ((S + 15) / 16) * GRU8X_WG_FLOATS;
      for (int e = 0; e < GRU8X_WG_FLOATS; ++e) { worst_wg = fmax(worst_wg, fabs(got[e] - part[e])); scale_wg = fmax(scale_wg, fabs(part[e])); }
    }
    printf("  k_gru8x_bwd (bf16x3, recompute) %8.2f us   tile 0 vs fp64 host: max |dX err| = %.3g (scale %.3g)   max |wg err| = %.3g (scale %.3g)\n",
           us, worst_dx, scale_dx, worst_wg, scale_wg);
  }
  // ---- backward 16 -> 16
  std::vector<float> refdX, refwg;
#define BWD(NAME, KERNEL, GRIDX)                                                                                      \
  {                                                                                                                   \
    CK(hipMemset(B.dX[0], 0, ndX * 4)); CK(hipMemset(B.wg[0], 0, nwg * 4));                                           \
    const float us = time_us([&] { hipLaunchKernelGGL(KERNEL, dim3(GRIDX, 2, 2), dim3(64), 0, 0, a16, b16, T); });   \
    if (refdX.empty()) { refdX = fetch(B.dX[0], ndX); refwg = fetch(B.wg[0], nwg); }                                  \
    printf("  %-36s %8.2f us   max |dX - first| = %.3g   max |wg - first| = %.3g\n", NAME, us, max_diff(B.dX[0], refdX, ndX), max_diff(B.wg[0], refwg, nwg)); \
  }
  BWD("k_gru16m_bwd (round 4, fp32 MFMA)", k_gru16m_bwd, dof_cdiv(S, 16))
  {
    CK(hipMemset(B.dX[0], 0, ndX * 4)); CK(hipMemset(B.wg[0], 0, nwg * 4));
    const float us = time_us([&] { hipLaunchKernelGGL((k_gru16x_bwd<true>), dim3(dof_cdiv(S, 64), 2, 2), dim3(256), 0, 0, a16, b16, T); });
    printf("  %-36s %8.2f us   max |dX - first| = %.3g   max |wg - first| = %.3g\n", "k_gru16x_bwd (bf16 x 3 pieces)", us, max_diff(B.dX[0], refdX, ndX), max_diff(B.wg[0], refwg, nwg));
  }
  return 0;
}
