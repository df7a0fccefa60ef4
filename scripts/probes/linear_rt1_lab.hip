// linear_rt1_lab.hip - does the K-resident Linear kernel gain from THREE waves per SIMD?  K = 384 with one 32-row tile per
// wave (96 operand registers, <= 170 per wave, two workgroups of six waves per CU) against the product (two 32-row tiles per
// wave, 256 registers, two waves per SIMD), same inputs, outputs compared.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDSS_LIN_LAB_RT1 scripts/probes/linear_rt1_lab.hip deep-spectral-segmentation_amd/csrc/lib.hip -o scripts/probes/linear_rt1_lab
#include "linear384_r4_lab.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1018 * 901, K = 384;
  struct Case { const char* name; int N, gelu, planar; } cases[] = {{"qkv", 1152, 0, 1}, {"proj", 384, 0, 1}, {"fc1+gelu", 1536, 1, 0}, {"fc1", 1536, 0, 0}};
  std::vector<_Float16> ha((size_t)M * K), hw((size_t)1536 * K), hb(1536);
  unsigned s = 99u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 16) % 2001 - 1000) * 0.001f; };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hw) v = (_Float16)(rnd() * 0.05f);
  for (auto& v : hb) v = (_Float16)(rnd() * 0.1f);
  _Float16 *A, *W, *B, *C, *C2;
  if (hipMalloc(&A, ha.size() * 2) != hipSuccess || hipMalloc(&W, hw.size() * 2) != hipSuccess || hipMalloc(&B, hb.size() * 2) != hipSuccess ||
      hipMalloc(&C, (size_t)M * 1536 * 2) != hipSuccess || hipMalloc(&C2, (size_t)M * 1536 * 2) != hipSuccess) return 1;
  (void)hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<_Float16> o1((size_t)4096 * 64), o2((size_t)4096 * 64);
  for (auto& c : cases) {
    float t[2];
    for (int v = 0; v < 2; ++v) {
      auto run = [&]() {
        if (v == 0) dss_linear_k384(A, W, B, C, M, c.N, c.gelu, c.planar, DSS_F16, nullptr);
        else dss_linear_k384_rt1(A, W, B, C2, M, c.N, c.gelu, c.planar, nullptr);
      };
      for (int w = 0; w < 3; ++w) run();
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) run();
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&t[v], e0, e1);
      t[v] *= 100.f;   // us per launch
    }
    // compare the first 256 K outputs and a slice near the end
    double maxd = 0, maxv = 0;
    for (size_t off : {(size_t)0, (size_t)M * c.N - o1.size()}) {
      (void)hipMemcpy(o1.data(), C + off, o1.size() * 2, hipMemcpyDeviceToHost);
      (void)hipMemcpy(o2.data(), C2 + off, o2.size() * 2, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < o1.size(); ++i) { maxd = fmax(maxd, fabs((double)o1[i] - (double)o2[i])); maxv = fmax(maxv, fabs((double)o1[i])); }
    }
    const double fl = 2.0 * M * c.N * K;
    printf("%-9s N=%4d M=%d: product (2 tiles/wave, 2 waves/SIMD) %7.1f us (%4.0f TF/s) | 1 tile/wave, 3 waves/SIMD %7.1f us (%4.0f TF/s) = %+5.1f %% | max |diff| %.2e (|out| max %.2f)\n",
           c.name, c.N, M, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, 100.0 * (t[1] / t[0] - 1), maxd, maxv);
  }
  return 0;
}
