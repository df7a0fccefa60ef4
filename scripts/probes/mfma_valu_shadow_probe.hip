// mfma_valu_shadow_probe.hip - what does ONE wave per SIMD pay for N VALU instructions issued behind every MFMA?
// Loop body: v_mfma_f32_32x32x16_f16 (two independent accumulator chains, alternating) followed by N instructions of one
// kind on registers the MFMA does not touch: v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_rcp_f32, v_accvgpr_read, v_cvt_pk.
// Printed: shader cycles per loop body (s_memtime), 256 workgroups of 256 threads (one wave per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_valu_shadow_probe.hip -o scripts/probes/mfma_valu_shadow_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int N, int WITH_MFMA>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, float seed) {
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = seed; acc1[i] = seed * 2; }
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = f32x2{seed + i, seed - i};
  f32x2 c = {seed * 0.5f, seed * 0.25f};
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 2000; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (WITH_MFMA) {
        if (h == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int n = 0; n < N; ++n) {
        f32x2& x = p[n & 7];
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c[0]), "v"(c[1]));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(c));
        if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
        if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[0]));
        if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[0]) : "v"(c[0]));
        if (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[0]) : "v"(c[0]), "v"(c[1]));
        if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
        if (KIND == 7) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c[0]), "v"(c[1]));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  if (s == 12345.678f) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND, int N, int M>
static double run(unsigned long long* d) {
  probe<KIND, N, M><<<256, 256>>>(d, 1.0f);
  probe<KIND, N, M><<<256, 256>>>(d, 1.0f);
  unsigned long long h = 0;
  (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  return (double)h / 4000.0;   // per (MFMA + N VALU)
}

template <int KIND>
static void row(const char* name, unsigned long long* d) {
  printf("%-16s alone N=8: %6.1f | behind an MFMA, N = 0, 2, 4, 6, 7, 8, 10, 12, 16: %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f cycles per MFMA\n", name,
         run<KIND, 8, 0>(d), run<KIND, 0, 1>(d), run<KIND, 2, 1>(d), run<KIND, 4, 1>(d), run<KIND, 6, 1>(d), run<KIND, 7, 1>(d), run<KIND, 8, 1>(d),
         run<KIND, 10, 1>(d), run<KIND, 12, 1>(d), run<KIND, 16, 1>(d));
}

int main() {
  unsigned long long* d;
  if (hipMalloc(&d, 16) != hipSuccess) return 1;
  row<0>("v_fma_f32", d);
  row<4>("v_mul_f32", d);
  row<1>("v_pk_fma_f32", d);
  row<2>("v_pk_mul_f32", d);
  row<6>("v_pk_add_f32", d);
  row<3>("v_rcp_f32", d);
  row<5>("v_cvt_pk_f16_f32", d);
  row<7>("v_pk_fma_f16", d);
  return 0;
}
