// attn_timeline_probe.hip - where does the wall time of the attention launch go?  Every workgroup of the product kernel
// records its start / end (s_memrealtime, 100 MHz) and the CU it ran on (HW_REG_HW_ID, HW_REG_XCC_ID); the host prints
// the concurrency per CU, the busy fraction of the CU slots and the dead time between consecutive workgroups of a slot.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee attn_timeline_probe.hip \
//        ../../deep-spectral-segmentation_amd/csrc/lib.hip -o attn_timeline_probe
#define DSS_ATTN_TIMELINE
#include "../../deep-spectral-segmentation_amd/csrc/attention.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 290, T = argc > 2 ? atoi(argv[2]) : 901, H = argc > 3 ? atoi(argv[3]) : 6;
  const size_t n = (size_t)B * T * 3 * H * 64;
  std::vector<_Float16> h(n);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.001f * 1.0f); }
  _Float16 *qkv, *out;
  const int nblocks = ((T + 255) / 256) * H * B;
  unsigned long long* tl;
  if (hipMalloc(&qkv, n * 2) != hipSuccess || hipMalloc(&out, n * 2 / 3) != hipSuccess ||
      hipMalloc(&tl, (size_t)nblocks * 32) != hipSuccess) return 1;
  (void)hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(dss_timeline_buf), &tl, sizeof(tl));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 4; ++it) {
    (void)hipEventRecord(e0);
    if (dss_attention_fwd(qkv, DSS_PLANAR64, out, B, T, H, 0.125f, DSS_F16, nullptr)) { printf("%s\n", dss_last_error()); return 1; }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> t((size_t)nblocks * 4);
  (void)hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;   // (xcc, se, cu) -> intervals
  double busy = 0;
  for (int b = 0; b < nblocks; ++b) {
    const unsigned long long a = t[4 * b], e = t[4 * b + 1];
    const unsigned hw = (unsigned)t[4 * b + 2], xcc = (unsigned)t[4 * b + 3] & 0xf;
    const unsigned key = (xcc << 16) | (((hw >> 13) & 0x7) << 8) | ((hw >> 8) & 0xf);   // gfx9 HW_ID: CU_ID [11:8], SE_ID [15:13]
    cu[key].push_back({a, e});
    t0 = std::min(t0, a); t1 = std::max(t1, e);
    busy += (double)(e - a);
  }
  const double span = (double)(t1 - t0);
  printf("B=%d T=%d H=%d: kernel %.1f us by events, %.1f us first start -> last end; %d workgroups on %zu distinct CUs\n", B, T, H,
         ms * 1e3, span / 100.0, nblocks, cu.size());
  printf("mean workgroup duration %.2f us; sum of durations / span = %.2f workgroups resident on average (%.2f per CU)\n",
         busy / nblocks / 100.0, busy / span, busy / span / cu.size());
  // per CU: max concurrency and gap statistics
  double gap_sum = 0; long gaps = 0; int maxc = 0; std::vector<double> durs;
  for (auto& kv : cu) {
    auto& v = kv.second;
    std::vector<std::pair<unsigned long long, int>> ev;
    for (auto& iv : v) { ev.push_back({iv.first, +1}); ev.push_back({iv.second, -1}); }
    std::sort(ev.begin(), ev.end());
    int c = 0; unsigned long long last_end = 0;
    for (auto& e : ev) {
      if (e.second < 0) { last_end = e.first; --c; }
      else { if (last_end && c < 2) { gap_sum += (double)(e.first - last_end); ++gaps; } ++c; maxc = std::max(maxc, c); }
    }
  }
  for (int b = 0; b < nblocks; ++b) durs.push_back((double)(t[4 * b + 1] - t[4 * b]) / 100.0);
  std::sort(durs.begin(), durs.end());
  printf("max workgroups concurrently on one CU: %d; mean refill gap (end of a workgroup -> start of the next on that CU while below 2): %.2f us over %ld refills\n",
         maxc, gaps ? gap_sum / gaps / 100.0 : 0.0, gaps);
  printf("workgroup duration percentiles (us): min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f\n", durs[0], durs[durs.size() / 10],
         durs[durs.size() / 2], durs[durs.size() * 9 / 10], durs.back());
  // start-time histogram of the first 2 ms in 50 us bins: how many workgroups are resident
  return 0;
}
