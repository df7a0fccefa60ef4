// How much does a DEPENDENT chain of v_mfma_f32_32x32x16_f16 (every MFMA accumulates into the previous one's result, the
// pattern of S^T = K.Q^T over the 4 k-steps of head dim 64) cost against independent chains?  One wave per SIMD (256
// threads), cycles per MFMA from s_memtime.   hipcc --offload-arch=gfx950 -O3 -o mfma_chain_probe mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256, 1) void probe(const f16x8* src, float* out, long long* cyc, int iters) {
  f16x8 a = src[threadIdx.x], b = src[256 + threadIdx.x];
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k % CHAINS], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  f16x8* src; float* out; long long* cyc;
  hipMalloc(&src, 512 * sizeof(f16x8)); hipMemset(src, 0, 512 * sizeof(f16x8));
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 1024 * sizeof(long long));
  const int iters = 2000;
  long long h[1024];
  auto report = [&](const char* name) {
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 1024 * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += (double)h[i];
    printf("%-28s %6.1f s_memtime ticks per MFMA (256 workgroups = one wave per SIMD)\n", name, m / 256 / (8.0 * iters));
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, src, out, cyc, iters); report("1 chain (all dependent)");
    hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, src, out, cyc, iters); report("2 chains alternating");
    hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256), 0, 0, src, out, cyc, iters); report("4 chains");
  }
  return 0;
}
