// clock_probe.hip - what shader clock does this GPU sustain under a dense MFMA / VALU load?
// s_memtime counts shader clocks, s_memrealtime counts the constant 100 MHz reference: their ratio over a ~1 ms
// busy loop is the sustained clock.  Build: hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int mode>
__global__ __launch_bounds__(256) void burn(unsigned long long* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
  const unsigned long long t0 = __builtin_readcyclecounter();      // s_memtime
  const unsigned long long r0 = wall_clock64();                    // s_memrealtime (100 MHz)
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    } else {
      v0 = __builtin_fmaf(v0, 1.0001f, v1); v1 = __builtin_fmaf(v1, 0.9999f, v2);
      v2 = __builtin_fmaf(v2, 1.0002f, v3); v3 = __builtin_fmaf(v3, 0.9998f, v0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float s = v0 + v1 + v2 + v3;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 1] = r1 - r0; out[3 * blockIdx.x + 2] = (unsigned long long)s; }
}
// mode 2: the operand pattern of a real K-resident GEMM: 48 distinct A fragments with random data, 3 rotating B
// fragments, two accumulator chains - same MFMA count per iteration as 12 iterations of mode 0
__global__ __launch_bounds__(256, 2) void burn_real(unsigned long long* out, const f16x8* src, int iters) {
  f16x8 a0[24], a1[24], f[3];
  for (int s = 0; s < 24; ++s) { a0[s] = src[(threadIdx.x * 51 + s) & 4095]; a1[s] = src[(threadIdx.x * 53 + s + 1000) & 4095]; }
  for (int s = 0; s < 3; ++s) f[s] = src[(threadIdx.x * 57 + s + 2000) & 4095];
  f32x16 c0 = {}, c1 = {};
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 24; ++s) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[s % 3], a0[s], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[s % 3], a1[s], c1, 0, 0, 0);
    }
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 1] = r1 - r0; out[3 * blockIdx.x + 2] = (unsigned long long)s; }
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_pk(f32x2 x) {      // A&S 7.1.28 on float2 (v_pk_* ops)
  f32x2 ax; ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
  const f32x2 z = ax * 0.70710678118654752f;
  f32x2 q = z * 0.0000430638f + 0.0002765672f;
  q = q * z + 0.0001520143f; q = q * z + 0.0092705272f; q = q * z + 0.0422820123f; q = q * z + 0.0705230784f;
  q = q * z + 1.0f; q = q * q; q = q * q; q = q * q; q = q * q;
  f32x2 r; r[0] = __builtin_amdgcn_rcpf(q[0]); r[1] = __builtin_amdgcn_rcpf(q[1]);
  return x * 0.5f + ax * 0.5f * (1.0f - r);
}
__device__ __forceinline__ float gelu_sc(float x) {      // same, scalar fp32 ops
  const float ax = fabsf(x), z = ax * 0.70710678118654752f;
  float q = __builtin_fmaf(z, 0.0000430638f, 0.0002765672f);
  q = __builtin_fmaf(q, z, 0.0001520143f); q = __builtin_fmaf(q, z, 0.0092705272f); q = __builtin_fmaf(q, z, 0.0422820123f);
  q = __builtin_fmaf(q, z, 0.0705230784f); q = __builtin_fmaf(q, z, 1.0f); q = q * q; q = q * q; q = q * q; q = q * q;
  const float r = __builtin_amdgcn_rcpf(q);
  return __builtin_fmaf(ax * 0.5f, 1.0f - r, x * 0.5f);
}
template <int mode>
__global__ __launch_bounds__(256) void burn_gelu(unsigned long long* out, float* sink, int iters) {
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = (threadIdx.x * 37 % 101) * 0.05f - 2.5f + i * 0.01f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) { f32x2 p = {v[i], v[i + 1]}; p = gelu_pk(p); v[i] = p[0] + 1.0f; v[i + 1] = p[1] - 1.0f; }
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = gelu_sc(v[i]) + ((i & 1) ? -1.0f : 1.0f);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += v[i];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
  {
    unsigned long long* d3; hipMalloc(&d3, 2048 * 8); float* sink; hipMalloc(&sink, 2048 * 256 * 4);
    unsigned long long h3;
    for (int mode = 0; mode < 2; ++mode)
      for (int blocks : {256, 512}) {
        const int iters = 2000;
        if (mode == 0) burn_gelu<0><<<blocks, 256>>>(d3, sink, iters); else burn_gelu<1><<<blocks, 256>>>(d3, sink, iters);
        hipDeviceSynchronize();
        hipMemcpy(&h3, d3, 8, hipMemcpyDeviceToHost);
        printf("gelu %s blocks=%d (%d wave/SIMD): %.1f cycles per 32 elements per wave (+32 adds)\n", mode == 0 ? "packed" : "scalar",
               blocks, blocks / 256, (double)h3 / iters);
      }
  }
  {
    f16x8* src; hipMalloc(&src, 4096 * 16);
    _Float16* hsrc = new _Float16[4096 * 8];
    unsigned x = 12345;
    for (int i = 0; i < 4096 * 8; ++i) { x = x * 1664525u + 1013904223u; hsrc[i] = (_Float16)(((int)(x >> 8) % 2001 - 1000) * 0.001f); }
    hipMemcpy(src, hsrc, 4096 * 16, hipMemcpyHostToDevice);
    unsigned long long* d2; hipMalloc(&d2, 3 * 2048 * 8);
    unsigned long long h2[3];
    for (int blocks : {256, 512}) {
      const int iters = 20000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      burn_real<<<blocks, 256>>>(d2, src, iters); hipDeviceSynchronize();
      hipEventRecord(e0); burn_real<<<blocks, 256>>>(d2, src, iters); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h2, d2, 24, hipMemcpyDeviceToHost);
      printf("real-operand mfma blocks=%4d: %.2f ms  shader clock %.0f MHz  cycles/48 mfma %.1f  = %.0f TF/s\n", blocks, ms,
             100.0 * (double)h2[0] / (double)h2[1], (double)h2[0] / iters, 48.0 * 32768 * iters * blocks * 4 / ms / 1e9);
    }
  }
  unsigned long long* d; hipMalloc(&d, 3 * 2048 * 8);
  unsigned long long h[6];
  for (int mode = 0; mode < 2; ++mode)
    for (int blocks : {1, 256, 512, 1024}) {
      const int iters = 40000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      if (mode == 0) burn<0><<<blocks, 256>>>(d, iters); else burn<1><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
      hipEventRecord(e0); if (mode == 0) burn<0><<<blocks, 256>>>(d, iters); else burn<1><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
      const double mhz = 100.0 * (double)h[0] / (double)h[1];
      const double flops = mode == 0 ? 4.0 * 32768 * iters * blocks * 4 : 0;
      printf("%s blocks=%4d: %.2f ms  shader clock %.0f MHz (memtime %llu / realtime %llu)  cycles/iter/wave %.1f  %s %.0f TF/s\n",
             mode == 0 ? "mfma" : "valu", blocks, ms, mhz, h[0], h[1], (double)h[0] / iters, mode == 0 ? "=" : "", flops / ms / 1e9);
    }
  return 0;
}
