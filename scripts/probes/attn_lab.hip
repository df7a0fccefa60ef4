// attn_lab.hip - A/B bench of the attention kernel and its ablation switches on the bench shapes (random data), interleaved
// rounds in ONE process; the product kernel is checked against an fp64 host reference on one image.  (Round 3's candidates -
// the round-2 kernel, 128-key stages, four-wave workgroups, a persistent grid, packed / scalar / log2-domain softmax - were
// compared with this harness before the losers were deleted: profiles/r03_attention_lab.txt.)  Build (scripts/gpu_r3.sh lab):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee scripts/probes/attn_lab.hip \
//         deep-spectral-segmentation_amd/csrc/lib.hip -o scripts/probes/attn_lab
// Run:   attn_lab [B T H planar [only-variant]]
#define DSS_ATTN_LAB
#include "../../deep-spectral-segmentation_amd/csrc/attention.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef void (*launch_fn)(const void*, void*, int, int, int, float, hipStream_t, int);
struct Variant { const char* name; launch_fn fn; bool exact; };
static const Variant VARIANTS[] = {
    {"attn_fwd (product)", dss::launch_attention<dss::f16, 0>, true},
    {"attn_fwd NO BARRIER", dss::launch_attention<dss::f16, 4>, false},
    {"attn_fwd NO DMA", dss::launch_attention<dss::f16, 8>, false},
    {"attn_fwd NO BARRIER NO DMA", dss::launch_attention<dss::f16, 12>, false},
};
static const int NV = sizeof(VARIANTS) / sizeof(VARIANTS[0]);

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 290, T = argc > 2 ? atoi(argv[2]) : 901, H = argc > 3 ? atoi(argv[3]) : 6;
  const int planar = argc > 4 ? atoi(argv[4]) : 1;
  const int only = argc > 5 ? atoi(argv[5]) : -1;
  const int rounds = argc > 6 ? atoi(argv[6]) : 5;
  const size_t nrow = (size_t)B * T, n = nrow * 3 * H * 64;
  // logical tensor qkv[b][t][3][h][64]; stored interleaved or planar [3h][B*T][64]
  std::vector<_Float16> h(n);
  std::vector<float> logical((size_t)T * 3 * H * 64);   // image 0 only, for the reference
  unsigned s = 12345u;
  for (size_t r = 0; r < nrow; ++r)
    for (int c = 0; c < 3 * H * 64; ++c) {
      s = s * 1664525u + 1013904223u;
      const float v = ((int)(s >> 16) % 2001 - 1000) * 0.001f * 1.5f;
      const _Float16 hv = (_Float16)v;
      if (planar) h[(size_t)(c / 64) * nrow * 64 + r * 64 + c % 64] = hv;
      else h[r * 3 * H * 64 + c] = hv;
      if (r < (size_t)T) logical[r * 3 * H * 64 + c] = (float)hv;
    }
  _Float16 *qkv, *out, *ref;
  const size_t nout = nrow * H * 64;
  if (hipMalloc(&qkv, n * 2) != hipSuccess || hipMalloc(&out, nout * 2) != hipSuccess ||
      hipMalloc(&ref, nout * 2) != hipSuccess) return 1;
  (void)hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const double flops = 4.0 * T * (double)T * H * 64 * B;
  printf("B=%d T=%d H=%d planar=%d  (%.3e FLOP per launch)\n", B, T, H, planar, flops);
  // reference output of the product kernel + fp64 check of image 0, head 0 and head H-1
  VARIANTS[0].fn(qkv, ref, B, T, H, 0.125f, nullptr, planar);
  if (hipDeviceSynchronize() != hipSuccess) { printf("product kernel failed\n"); return 1; }
  std::vector<_Float16> href(nout), hout(nout);
  (void)hipMemcpy(href.data(), ref, nout * 2, hipMemcpyDeviceToHost);
  {
    double maxerr = 0;
    for (int hd : {0, H - 1}) {
      std::vector<double> p(T);
      for (int q = 0; q < T; q += 7) {
        double mx = -1e300;
        for (int k = 0; k < T; ++k) {
          double d = 0;
          for (int e = 0; e < 64; ++e)
            d += (double)logical[(size_t)q * 3 * H * 64 + hd * 64 + e] * logical[(size_t)k * 3 * H * 64 + (H + hd) * 64 + e];
          p[k] = d * 0.125; mx = std::max(mx, p[k]);
        }
        double sum = 0;
        for (int k = 0; k < T; ++k) { p[k] = std::exp(p[k] - mx); sum += p[k]; }
        for (int e = 0; e < 64; ++e) {
          double o = 0;
          for (int k = 0; k < T; ++k) o += p[k] * logical[(size_t)k * 3 * H * 64 + (2 * H + hd) * 64 + e];
          maxerr = std::max(maxerr, std::fabs(o / sum - (double)href[(size_t)q * H * 64 + hd * 64 + e]));
        }
      }
    }
    printf("product kernel vs fp64 (image 0, heads 0 and %d, every 7th query): max abs err %.3e\n", H - 1, maxerr);
  }
  std::vector<std::vector<float>> times(NV);
  for (int rnd = 0; rnd < rounds; ++rnd)
    for (int v = 0; v < NV; ++v) {
      if (only >= 0 && v != only) continue;
      (void)hipMemsetAsync(out, 0xff, nout * 2);
      VARIANTS[v].fn(qkv, out, B, T, H, 0.125f, nullptr, planar);   // warm + result
      if (rnd == 0) {
        if (hipDeviceSynchronize() != hipSuccess) { printf("variant %d FAILED to run\n", v); return 1; }
        (void)hipMemcpy(hout.data(), out, nout * 2, hipMemcpyDeviceToHost);
        size_t diff = 0; double maxd = 0;
        for (size_t i = 0; i < nout; ++i) {
          if (memcmp(&hout[i], &href[i], 2) != 0) ++diff;
          const double d = std::fabs((double)hout[i] - (double)href[i]);
          if (!(d <= maxd)) maxd = d;
        }
        printf("  [%d] %-34s vs product: %zu of %zu values differ, max |d| %.3e %s\n", v, VARIANTS[v].name, diff, nout, maxd,
               VARIANTS[v].exact ? (diff ? "<-- MISMATCH" : "(bit-identical)") : (maxd <= 2e-3 ? "(close)" : "(ablation / not expected to match)"));
      }
      (void)hipEventRecord(e0);
      for (int it = 0; it < 5; ++it) VARIANTS[v].fn(qkv, out, B, T, H, 0.125f, nullptr, planar);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      times[v].push_back(ms / 5);
    }
#ifdef DSS_ATTN_CLOCK
  {   // sustained shader clock inside each variant's workgroups
    int wc_khz = 0;
    (void)hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    for (int v = 0; v < NV; ++v) {
      if (only >= 0 && v != only) continue;
      unsigned long long z[4] = {0, 0, 0, 0}, r[4];
      (void)hipMemcpyToSymbol(HIP_SYMBOL(dss_clock_buf), z, sizeof(z));
      for (int it = 0; it < 3; ++it) VARIANTS[v].fn(qkv, out, B, T, H, 0.125f, nullptr, planar);
      (void)hipDeviceSynchronize();
      (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(dss_clock_buf), sizeof(r));
      printf("clock %-34s %llu workgroups: mean %.0f shader cycles in %.2f us each -> %.0f MHz sustained (wall clock %d kHz)\n",
             VARIANTS[v].name, r[2], (double)r[0] / r[2], (double)r[1] / r[2] / wc_khz * 1e3, (double)r[0] / r[1] * wc_khz * 1e-3, wc_khz);
    }
  }
#endif
  for (int v = 0; v < NV; ++v) {
    if (times[v].empty()) continue;
    std::sort(times[v].begin(), times[v].end());
    const float mn = times[v][0], md = times[v][times[v].size() / 2];
    printf("%-36s min %7.1f us  median %7.1f us  -> %6.0f TF/s (min)  %6.0f (median)\n", VARIANTS[v].name, mn * 1e3, md * 1e3,
           flops / mn / 1e9, flops / md / 1e9);
  }
  return 0;
}
