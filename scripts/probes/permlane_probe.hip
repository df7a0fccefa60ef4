// Probe: __builtin_amdgcn_permlane32_swap return convention on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
  const unsigned l = threadIdx.x;
  const unsigned a = 100 + l, b = 200 + l;
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l * 4 + 0] = r[0]; out[l * 4 + 1] = r[1];
  const auto q = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  out[l * 4 + 2] = q[0]; out[l * 4 + 3] = q[1];
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 64 * 16);
  unsigned h[256];
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 7) printf("lane %2d: swap(a=100+l,b=200+l) -> r0=%u r1=%u | swap(a,a) -> %u %u\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  return 0;
}
