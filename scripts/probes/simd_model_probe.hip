// simd_model_probe.hip - how do MFMA and VALU work of SEVERAL waves share one SIMD?  Every wave runs the instruction
// pattern of the attention inner loop on registers only:
//     4 dependent v_mfma_f32_32x32x16_f16 (scores)  ->  NV VALU instructions that read the scores  ->  8 v_cvt_pk
//     ->  4 v_mfma (two accumulator chains, the P.V product)
// with W = 1..4 waves per SIMD (one workgroup of 256 W threads per CU; 100 KB of LDS keeps a second one out).  Printed:
// shader cycles per iteration (completion time of the slowest of the SIMD's W waves / iterations) for every (W, NV, kind); kind = fma / exp / add / attention mix.  PATTERN 1 is the
// same work software-pipelined INSIDE the wave: the score MFMAs of the next tile are interleaved with the VALU work of
// the current one (1 MFMA, then NV/4 VALU, ...), the instruction stream the compiler does not produce.
// Build: hipcc --offload-arch=gfx950 -O3 simd_model_probe.hip -o simd_model_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA0(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))

template <int KIND>
__device__ __forceinline__ void valu_op(f32x16& s, int i, float& acc, float c, float d) {
  float x = s[i];
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
  if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x));
  if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));          // independent adds (no chain)
  if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
  if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(c));
  if (KIND == 10) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x) : "v"(c), "v"(d));  // no read of an MFMA result at all
  s[i] = x;
}
// the attention mix on 16 scores: NV = 48 -> 16 fma + 16 exp + 16 add; NV = 32 -> 16 exp + 16 add; NV = 16 -> 16 exp
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pk_add(f32x16& s, int i, float (&acc)[4], int j) {   // acc[j..j+1] += s[i..i+1]
  f32x2 a = {acc[j], acc[j + 1]}, x = {s[i], s[i + 1]};
  asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(x));
  acc[j] = a[0]; acc[j + 1] = a[1];
}
__device__ __forceinline__ void pk_fma(f32x16& s, int i, float c, float d) {
  f32x2 x = {s[i], s[i + 1]}, cc = {c, c}, dd = {d, d};
  asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(cc), "v"(dd));
  s[i] = x[0]; s[i + 1] = x[1];
}
template <int NV, int KIND>
__device__ __forceinline__ void valu_block(f32x16& s, float (&acc)[4], float c, float d, int from, int to, float (&acc8)[8]) {
  if (KIND >= 11 && KIND <= 16) {
    // NV = 32 instructions: 16 exp + 16 add in different orders / dependency shapes
    constexpr int LAG = KIND == 11 ? 4 : (KIND == 12 ? 8 : 16);
    int n = 0;
#pragma unroll
    for (int i = 0; i < 16 + LAG; ++i) {
      if (i < 16) { if (n >= from && n < to) valu_op<1>(s, i, acc[0], c, d); ++n; }
      if (i >= LAG && i - LAG < 16) {
        const int j = i - LAG;
        if (n >= from && n < to) {
          float x = s[j];
          if (KIND == 13) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[0]) : "v"(x));
          else if (KIND == 14) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc8[j & 7]) : "v"(x)); }
          else if (KIND == 15) { if (j & 1) { float y = s[j - 1]; asm volatile("v_add_f32 %0, %0, %1" : "+v"(y) : "v"(x)); s[j - 1] = y; } else asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[(j >> 1) & 3]) : "v"(x)); }
          else if (KIND == 16) asm volatile("v_fma_f32 %0, %1, 1.0, %0" : "+v"(acc[j & 3]) : "v"(x));
          else asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[j & 3]) : "v"(x));
        }
        ++n;
      }
    }
    return;
  }
  if (KIND >= 4) {
    // op list: [16 fma | 8 pk_fma | none] then 16 exp then 8 pk_add
    constexpr int NF = KIND == 6 ? 16 : (KIND == 5 ? 8 : 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < from || i >= to) continue;
      if (i < NF) { if (KIND == 6) valu_op<0>(s, i, acc[0], c, d); else pk_fma(s, 2 * i, c, d); }
      else if (i < NF + 16) valu_op<1>(s, i - NF, acc[0], c, d);
      else pk_add(s, 2 * (i - NF - 16), acc, 2 * ((i - NF - 16) & 1));
    }
    return;
  }
  // ops are numbered 0 .. NV-1; executes ops [from, to)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < from || i >= to) continue;
    if (KIND == 3) {
      const int phase = NV == 48 ? i / 16 : (NV == 32 ? 1 + i / 16 : 1);
      if (phase == 0) valu_op<0>(s, i % 16, acc[0], c, d);
      if (phase == 1) valu_op<1>(s, i % 16, acc[0], c, d);
      if (phase == 2) valu_op<2>(s, i % 16, acc[i % 4], c, d);
    } else {
      valu_op<KIND>(s, i % 16, acc[i % 4], c, d);
    }
  }
}

template <int NV, int KIND, int PATTERN>
__global__ __launch_bounds__(1024) void probe(unsigned long long* out, int iters, float c, float d) {
  extern __shared__ char lds_keepout[];
  f16x8 k[4], q[4], v[4];
  unsigned pk[8];
  const float seed = (threadIdx.x % 13) * 0.01f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { k[i][e] = (_Float16)(seed + 0.01f * e); q[i][e] = (_Float16)(0.02f * i - seed); v[i][e] = (_Float16)(0.5f - seed); }
  f32x16 s, s2, o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; s2[r] = seed; s[r] = seed; }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (PATTERN == 2 || PATTERN == 3) {   // log2-domain softmax with the row sums taken from the PACKED probabilities (v_dot2_f32_f16)
      MFMA0(s, k[0], q[0]); MFMA(s, k[1], q[1]); MFMA(s, k[2], q[2]); MFMA(s, k[3], q[3]);
      asm volatile("s_nop 15" ::: "memory");
      valu_block<16, 1>(s, acc, c, d, 0, 16, acc8);
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(s[2 * i]), "v"(s[2 * i + 1]));
      const unsigned ones = 0x3c003c00u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (PATTERN == 2) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(pk[i]), "v"(ones));
        else asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc[i & 3]) : "v"(pk[i]), "v"(ones));
      }
      f16x8 p0, p1;
      { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 a = {pk[0], pk[1], pk[2], pk[3]}, b = {pk[4], pk[5], pk[6], pk[7]};
        p0 = __builtin_bit_cast(f16x8, a); p1 = __builtin_bit_cast(f16x8, b); }
      asm volatile("s_nop 1" ::: "memory");
      MFMA(o0, v[0], p0); MFMA(o1, v[1], p0); MFMA(o0, v[2], p1); MFMA(o1, v[3], p1);
    } else if (PATTERN == 0) {
      MFMA0(s, k[0], q[0]); MFMA(s, k[1], q[1]); MFMA(s, k[2], q[2]); MFMA(s, k[3], q[3]);
      asm volatile("s_nop 15" ::: "memory");
      valu_block<NV, KIND>(s, acc, c, d, 0, NV, acc8);
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(s[2 * i]), "v"(s[2 * i + 1]));
      f16x8 p0, p1;
      { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 a = {pk[0], pk[1], pk[2], pk[3]}, b = {pk[4], pk[5], pk[6], pk[7]};
        p0 = __builtin_bit_cast(f16x8, a); p1 = __builtin_bit_cast(f16x8, b); }
      asm volatile("s_nop 1" ::: "memory");
      MFMA(o0, v[0], p0); MFMA(o1, v[1], p0); MFMA(o0, v[2], p1); MFMA(o1, v[3], p1);
    } else {
      // software pipelined inside the wave: scores of the NEXT tile between the VALU ops on the CURRENT tile;
      // two steps per loop trip with the roles of s / s2 swapped (no register copies)
      constexpr int Q4 = NV / 4;
#define PIPE_STEP(SC, SN)                                                                                   \
      MFMA0(SN, k[0], q[0]); valu_block<NV, KIND>(SC, acc, c, d, 0, Q4, acc8);                                     \
      MFMA(SN, k[1], q[1]);  valu_block<NV, KIND>(SC, acc, c, d, Q4, 2 * Q4, acc8);                                \
      MFMA(SN, k[2], q[2]);  valu_block<NV, KIND>(SC, acc, c, d, 2 * Q4, 3 * Q4, acc8);                            \
      MFMA(SN, k[3], q[3]);  valu_block<NV, KIND>(SC, acc, c, d, 3 * Q4, NV, acc8);                                \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                         \
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(SC[2 * i]), "v"(SC[2 * i + 1]));      \
      { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 a = {pk[0], pk[1], pk[2], pk[3]}, b = {pk[4], pk[5], pk[6], pk[7]}; \
        f16x8 p0 = __builtin_bit_cast(f16x8, a), p1 = __builtin_bit_cast(f16x8, b);                         \
        asm volatile("s_nop 1" ::: "memory");                                                               \
        MFMA(o0, v[0], p0); MFMA(o1, v[1], p0); MFMA(o0, v[2], p1); MFMA(o1, v[3], p1); }
      PIPE_STEP(s, s2)
      if (++it >= iters) break;
      PIPE_STEP(s2, s)
    }
  }
  asm volatile("s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += acc8[i];
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += o0[r] + o1[r] + s[r] + s2[r];
  if (sum == 123.456f) out[1000000] = 1;   // keep everything live
  const unsigned long long r1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (threadIdx.x == 0) out[8192 + blockIdx.x] = r1 - r0;
}

static double g_ms = 0, g_mhz = 0;
template <int NV, int KIND, int PATTERN>
static double run(int W, unsigned long long* dbuf, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)probe<NV, KIND, PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NV, KIND, PATTERN>), dim3(256), dim3(256 * W), 100 * 1024, 0, dbuf, iters, 1.0001f, 0.001f);
    (void)hipEventRecord(e1);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 0; }
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); g_ms = ms;
  std::vector<unsigned long long> h(8192 + 256);
  (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
  // a SIMD's W waves start together; oldest-first arbitration lets them FINISH at different times, so the cost of an
  // iteration is the completion time of the slowest wave (mean over blocks of the max over the block's waves)
  double sum = 0; int n = 0;
  for (int b = 0; b < 256; ++b) {
    unsigned long long mx = 0;
    for (int w = 0; w < 4 * W; ++w) mx = h[b * 16 + w] > mx ? h[b * 16 + w] : mx;
    sum += (double)mx; ++n;
  }
  double rt = 0;
  for (int b = 0; b < 256; ++b) rt += (double)h[8192 + b];
  g_mhz = sum / n / (rt / 256) * 100.0;
  return sum / n / iters;
}

template <int NV, int KIND, int PATTERN>
static void row(const char* name, unsigned long long* dbuf) {
  printf("%-44s", name);
  for (int W = 1; W <= 4; ++W) {
    const double c = run<NV, KIND, PATTERN>(W, dbuf, 4000);
    printf("  W=%d: %5.0f (%4.0f/wave) %4.0fTF", W, c, c / W, 4000.0 * 8 * 32768 * 4 * W * 256 / (g_ms * 1e-3) / 1e12);
  }
  printf("\n");
}

int main() {
  unsigned long long* dbuf;
  if (hipMalloc(&dbuf, (1000000 + 16) * 8) != hipSuccess) return 1;
  printf("cycles per iteration (8 MFMA 32x32x16 = 256 cycles of matrix pipe + 8 v_cvt_pk + NV VALU), per wave and per SIMD (per wave / W)\n");
  row<0, 0, 0>("serial, NV=0 (MFMA + 8 cvt only)", dbuf);
  row<16, 0, 0>("serial, 16 v_fma", dbuf);
  row<32, 0, 0>("serial, 32 v_fma", dbuf);
  row<48, 0, 0>("serial, 48 v_fma", dbuf);
  row<64, 0, 0>("serial, 64 v_fma", dbuf);
  row<16, 1, 0>("serial, 16 v_exp", dbuf);
  row<32, 1, 0>("serial, 32 v_exp", dbuf);
  row<16, 2, 0>("serial, 16 v_add (4 chains)", dbuf);
  row<32, 2, 0>("serial, 32 v_add (4 chains)", dbuf);
  row<16, 3, 0>("serial, attention mix 16 (exp)", dbuf);
  row<32, 3, 0>("serial, attention mix 32 (exp+add)", dbuf);
  row<48, 3, 0>("serial, attention mix 48 (fma+exp+add)", dbuf);
  row<32, 11, 0>("serial, 16 exp + 16 add interleaved (lag 4)", dbuf);
  row<32, 12, 0>("serial, 16 exp + 16 add interleaved (lag 8)", dbuf);
  row<32, 13, 0>("serial, 16 exp then 16 add ONE chain", dbuf);
  row<32, 14, 0>("serial, 16 exp then 16 add 8 chains", dbuf);
  row<32, 15, 0>("serial, 16 exp then 16 add pairwise tree", dbuf);
  row<32, 16, 0>("serial, 16 exp then 16 v_fma(x,1,acc) 4 chains", dbuf);
  row<16, 7, 0>("serial, 16 independent v_add (x += c)", dbuf);
  row<32, 7, 0>("serial, 32 independent v_add (x += c)", dbuf);
  row<16, 8, 0>("serial, 16 v_mul", dbuf);
  row<16, 9, 0>("serial, 16 v_mov", dbuf);
  row<16, 10, 0>("serial, 16 v_add of non-MFMA registers", dbuf);
  row<32, 10, 0>("serial, 32 v_add of non-MFMA registers", dbuf);
  row<16, 1, 2>("serial, 16 exp + 8 v_dot2_f32_f16 (VOP3P)", dbuf);
  row<16, 1, 3>("serial, 16 exp + 8 v_dot2c_f32_f16", dbuf);
  row<24, 4, 0>("serial, 16 exp + 8 pk_add", dbuf);
  row<32, 5, 0>("serial, 8 pk_fma + 16 exp + 8 pk_add", dbuf);
  row<40, 6, 0>("serial, 16 fma + 16 exp + 8 pk_add", dbuf);
  row<24, 4, 1>("in-wave pipelined, 16 exp + 8 pk_add", dbuf);
  row<16, 3, 1>("in-wave pipelined, mix 16", dbuf);
  row<32, 3, 1>("in-wave pipelined, mix 32", dbuf);
  row<48, 3, 1>("in-wave pipelined, mix 48", dbuf);
  row<48, 0, 1>("in-wave pipelined, 48 v_fma", dbuf);
  return 0;
}
