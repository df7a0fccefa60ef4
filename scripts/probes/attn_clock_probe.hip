// attn_clock_probe.hip - time the product attention kernel stand-alone and read the shader clock the chip sustains INSIDE
// it (s_memtime = shader cycles vs s_memrealtime = 100 MHz, one workgroup in the middle of the grid).  Includes the
// product source with DSS_ATTN_CLOCK defined; the library itself carries no probe code.
// Build (cross-compiles without a GPU):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee attn_clock_probe.hip \
//         ../../deep-spectral-segmentation_amd/csrc/lib.hip -o attn_clock_probe
#define DSS_ATTN_CLOCK
#ifndef DSS_PROBE_BLOCK
#define DSS_PROBE_BLOCK 1600
#endif
#include "../../deep-spectral-segmentation_amd/csrc/attention.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 290, T = argc > 2 ? atoi(argv[2]) : 901, H = argc > 3 ? atoi(argv[3]) : 6;
  const size_t n = (size_t)B * T * 3 * H * 64;
  std::vector<_Float16> h(n);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.001f * 1.7f); }
  _Float16 *qkv, *out;
  if (hipMalloc(&qkv, n * 2) != hipSuccess || hipMalloc(&out, n * 2 / 3) != hipSuccess) return 1;
  (void)hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 8; ++it) {
    (void)hipEventRecord(e0);
    const int rc = dss_attention_fwd(qkv, DSS_PLANAR64, out, B, T, H, 0.125f, DSS_F16, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    if (rc) { printf("launch failed: %s\n", dss_last_error()); return 1; }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (it > 1 && ms < best) best = ms;
  }
  const double flops = 4.0 * T * T * H * 64.0 * B;
  unsigned long long c[4];
  (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(dss_clock_buf), sizeof(c));
  printf("B=%d T=%d H=%d: attn_fwd4 %.1f us (%.0f TF/s); workgroup %d: %llu shader cycles in %.1f us -> %.0f MHz sustained\n", B, T, H,
         best * 1e3, flops / best / 1e9, DSS_PROBE_BLOCK, c[0], c[1] / 100.0, c[1] ? 100.0 * c[0] / c[1] : 0.0);
  return 0;
}
