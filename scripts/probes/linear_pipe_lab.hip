// linear_pipe_lab.hip - can ONE wave hide a chunk's epilogue (fc1's GELU) under its own next chunk's MFMAs?  K = 384, the lab build
// of linear384.hip (-DDSS_LIN_LAB_PIPE): one workgroup of four waves per CU (one wave per SIMD, 512 registers: operand fragments
// in the accumulation registers unless -DDSS_LIN_LAB_PIPE_NO_AGPR), two accumulator sets, the epilogue of chunk c - 1 in ten stages
// behind the MFMAs of chunk c, against the product's kernel on the same inputs, product and lab alternating three times.
// Answer (profiles/r04_linear_lab.txt item 8): +8 % SLOWER for fc1 + GELU, +17 % plain.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDSS_LIN_LAB_PIPE=230 -DDSS_LIN_LAB_PIPE_RT=2 -DDSS_LIN_LAB_PIPE_PF=3
//        scripts/probes/linear_pipe_lab.hip deep-spectral-segmentation_amd/csrc/lib.hip -o scripts/probes/linear_pipe_lab_r2p3n0
//        (DSS_LIN_LAB_PIPE's value is only printed; _RT = row tiles per wave, _PF = W fragments read ahead)
#include "linear384_r4_lab.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

int main(int argc, char** argv) {
  const int M = (argc > 1 ? atoi(argv[1]) : 290 * 901) / 128 * 128, K = 384;   // the lab path has no ragged-block code
  struct Case { const char* name; int N, gelu; } cases[] = {{"fc1+gelu", 1536, 1}, {"fc1", 1536, 0}, {"proj", 384, 0}};
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
  std::vector<_Float16> o1((size_t)M * 64), o2((size_t)M * 64);
  for (auto& c : cases) {
    float t[2] = {1e30f, 1e30f}, tt[2][3];
    for (int rep = 0; rep < 3; ++rep)               // product, lab, product, lab, ...: the clock a kernel finds depends on what ran before it
      for (int v = 0; v < 2; ++v) {
        auto run = [&]() {
          if (v == 0) dss_linear_k384(A, W, B, C, M, c.N, c.gelu, 0, DSS_F16, nullptr);
          else dss_linear_k384_pipe(A, W, B, C2, M, c.N, c.gelu, nullptr);
        };
        for (int w = 0; w < 3; ++w) run();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) run();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&tt[v][rep], e0, e1);
        tt[v][rep] *= 50.f;   // us per launch
        t[v] = fminf(t[v], tt[v][rep]);
      }
    printf("   rounds: product %.1f %.1f %.1f | lab %.1f %.1f %.1f us\n", tt[0][0], tt[0][1], tt[0][2], tt[1][0], tt[1][1], tt[1][2]);
    double maxd = 0, maxv = 0;
    size_t bad = 0;
    for (size_t off = 0; off + o1.size() <= (size_t)M * c.N; off += o1.size() * 5) {   // every fifth 64-column-equivalent slab, first and last included below
      (void)hipMemcpy(o1.data(), C + off, o1.size() * 2, hipMemcpyDeviceToHost);
      (void)hipMemcpy(o2.data(), C2 + off, o2.size() * 2, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < o1.size(); ++i) {
        const double d = fabs((double)o1[i] - (double)o2[i]);
        if (!(d <= 2e-3)) ++bad;
        maxd = fmax(maxd, d); maxv = fmax(maxv, fabs((double)o1[i]));
      }
    }
    (void)hipMemcpy(o1.data(), C + (size_t)M * c.N - o1.size(), o1.size() * 2, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o2.data(), C2 + (size_t)M * c.N - o2.size(), o2.size() * 2, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < o1.size(); ++i) { const double d = fabs((double)o1[i] - (double)o2[i]); if (!(d <= 2e-3)) ++bad; maxd = fmax(maxd, d); }
    const double fl = 2.0 * M * c.N * K;
    printf("pipe=%d %-9s N=%4d M=%d: product %7.1f us (%4.0f TF/s) | one wave/SIMD, epilogue inside the next chunk's MFMAs %7.1f us (%4.0f TF/s) = %+5.1f %% | max |diff| %.2e, %zu outside 2e-3 (|out| max %.2f)\n",
           DSS_LIN_LAB_PIPE, c.name, c.N, M, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, 100.0 * (t[1] / t[0] - 1), maxd, bad, maxv);
  }
  return 0;
}
