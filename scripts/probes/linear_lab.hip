// linear_lab.hip - where does a workgroup of the K-resident Linear kernel spend its cycles?  The library kernel with
// DSS_LIN_TIMELINE: wave 0 of every workgroup accumulates s_memtime differences for the A prologue, the MFMA phases, the
// epilogues and the end-of-chunk wait + barrier.  Random f16 data (clock and power as in the bench).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/linear_lab.hip deep-spectral-segmentation_amd/csrc/lib.hip \
//        -o scripts/probes/linear_lab
#define DSS_LIN_TIMELINE
#include "../../deep-spectral-segmentation_amd/csrc/linear384.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 290 * 901, K = 384;
  struct Case { const char* name; int N, gelu, planar; } cases[] = {{"qkv", 1152, 0, 1}, {"proj", 384, 0, 1}, {"fc1+gelu", 1536, 1, 0}, {"fc1", 1536, 0, 0}};
  std::vector<_Float16> ha((size_t)M * K), hw((size_t)1536 * K), hb(1536);
  unsigned s = 99u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 16) % 2001 - 1000) * 0.001f; };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hw) v = (_Float16)(rnd() * 0.05f);
  for (auto& v : hb) v = (_Float16)(rnd() * 0.1f);
  _Float16 *A, *W, *B, *C;
  if (hipMalloc(&A, ha.size() * 2) != hipSuccess || hipMalloc(&W, hw.size() * 2) != hipSuccess || hipMalloc(&B, hb.size() * 2) != hipSuccess ||
      hipMalloc(&C, (size_t)M * 1536 * 2) != hipSuccess) return 1;
  (void)hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (auto& c : cases) {
    for (int w = 0; w < 3; ++w) dss_linear_k384(A, W, B, C, M, c.N, c.gelu, c.planar, DSS_F16, nullptr);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r[8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dss_lin_tl), z, sizeof(z));
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) dss_linear_k384(A, W, B, C, M, c.N, c.gelu, c.planar, DSS_F16, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(dss_lin_tl), sizeof(r));
    const double wg = (double)r[4], tot = (double)(r[0] + r[1] + r[2] + r[3]) / wg;
    printf("%-9s N=%4d: %7.1f us (%5.0f TF/s); per workgroup %8.0f cycles: A prologue %5.1f %%, MFMA phases %5.1f %%, epilogues %5.1f %%, "
           "wait+barrier %5.1f %%  (%.0f / %.0f / %.0f / %.0f cycles per chunk); workgroup lifetime %.1f us -> %.0f MHz; %.0f workgroups x lifetime / (512 slots x kernel time) = %.2f\n", c.name, c.N, ms / reps * 1e3,
           2.0 * M * c.N * K / (ms / reps * 1e-3) / 1e12, tot, 100.0 * r[0] / wg / tot, 100.0 * r[1] / wg / tot, 100.0 * r[2] / wg / tot,
           100.0 * r[3] / wg / tot, (double)r[0] / wg, (double)r[1] / wg / (c.N / 32), (double)r[2] / wg / (c.N / 32), (double)r[3] / wg / (c.N / 32),
           (double)r[5] / wg / 100.0, tot / ((double)r[5] / wg / 100.0), wg / reps, (wg / reps) * ((double)r[5] / wg / 100.0) / (512.0 * ms / reps * 1e3));
  }
  // ---- round 4: the LayerNorm-prologue variants (dss_lnlinear_k384): x f32 (+ residual f16, row-major) -> same outputs ----
  float *X, *AUX;
  if (hipMalloc(&X, (size_t)M * K * 4) != hipSuccess || hipMalloc(&AUX, 1536 * 2 * 4) != hipSuccess) return 1;
  {
    std::vector<float> hx((size_t)M * K), haux(1536 * 2);
    for (auto& v : hx) v = rnd() * 3.f;
    for (auto& v : haux) v = rnd();
    (void)hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(AUX, haux.data(), haux.size() * 4, hipMemcpyHostToDevice);
  }
  struct LCase { const char* name; int N, gelu, planar, res; } lcases[] = {{"LN+qkv (no residual)", 1152, 0, 1, 0}, {"res+LN+qkv", 1152, 0, 1, 1},
                                                                          {"LN+fc1+gelu (no res)", 1536, 1, 0, 0}, {"res+LN+fc1+gelu", 1536, 1, 0, 1}};
  for (auto& c : lcases) {
    for (int w = 0; w < 3; ++w) dss_lnlinear_k384(X, c.res ? A : nullptr, DSS_ROW_MAJOR, 1e-6f, W, AUX, C, M, c.N, c.gelu, c.planar, DSS_F16, nullptr);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r[8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dss_lin_tl), z, sizeof(z));
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) dss_lnlinear_k384(X, c.res ? A : nullptr, DSS_ROW_MAJOR, 1e-6f, W, AUX, C, M, c.N, c.gelu, c.planar, DSS_F16, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(dss_lin_tl), sizeof(r));
    const double wg = (double)r[4], tot = (double)(r[0] + r[1] + r[2] + r[3]) / wg;
    const double life_us = (double)r[5] / wg / 100.0;
    printf("%-22s N=%4d: %7.1f us; per workgroup %8.0f cycles: LN prologue %5.1f %% (%.0f cycles = %.1f us), MFMA phases %5.1f %%, epilogues %5.1f %%, "
           "wait+barrier %5.1f %%; workgroup lifetime %.1f us -> %.0f MHz; slot occupancy %.2f\n", c.name, c.N, ms / reps * 1e3, tot,
           100.0 * r[0] / wg / tot, (double)r[0] / wg, (double)r[0] / wg / (tot / life_us), 100.0 * r[1] / wg / tot, 100.0 * r[2] / wg / tot, 100.0 * r[3] / wg / tot,
           life_us, tot / life_us, (wg / reps) * life_us / (512.0 * ms / reps * 1e3));
  }
  return 0;
}
