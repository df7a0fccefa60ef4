// overlap_probe.hip - can the matrix pipe and the VALU of ONE SIMD be kept busy by two different waves at once?
// A 512-thread workgroup puts waves w and w + 4 on the same SIMD.  Waves 0-3 (group X) run ITER x 16 back-to-back
// v_mfma_f32_32x32x16_f16 (the accumulator pattern of the attention kernel's matrix phase), waves 4-7 (group Y) run
// ITER x a softmax-like VALU block.  Each group is timed alone (the other group exits at once) and together, with and
// without an s_barrier per iteration.  The VALU mix is a template bitmask so single instruction classes can be priced:
//   1 v_fma_f32 x32   2 v_exp_f32 x32   4 v_add_f32 x32 (one dependent chain)   8 v_cvt_pk_f16_f32 x16
//   16 v_max3_f32 x10 + v_permlane32_swap x2
// Build: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MIX>
__device__ __forceinline__ void valu_block(float (&x)[32], float& acc, unsigned (&p)[16], float c, float d) {
  if (MIX & 1) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
  }
  if (MIX & 2) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
  }
  if (MIX & 4) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x[i]));
  }
  if (MIX & 8) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[i]) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
  }
  if (MIX & 16) {
    float m0 = x[0], m1 = x[1];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(x[2 + 4 * i]), "v"(x[3 + 4 * i]));
      asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m1) : "v"(x[4 + 4 * i]), "v"(x[5 + 4 * i]));
    }
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(m0), "+v"(m1));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(m0), "+v"(m1));
    acc += m0 * 1e-30f + m1 * 1e-30f;
  }
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if (MIX & 32) {   // 16 x v_pk_fma_f32 (two floats per lane each)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      f32x2 v = {x[2 * i], x[2 * i + 1]}, cc = {c, c}, dd = {d, d};
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(cc), "v"(dd));
      x[2 * i] = v[0]; x[2 * i + 1] = v[1];
    }
  }
  if (MIX & 64) {   // 16 x v_pk_add_f32 into two independent float2 accumulators
    f32x2 a0 = {acc, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x2 u = {x[4 * i], x[4 * i + 1]}, w = {x[4 * i + 2], x[4 * i + 3]};
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a0) : "v"(u));
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a1) : "v"(w));
    }
    acc = a0[0] + a0[1] + a1[0] + a1[1];
  }
  if (MIX & 128) {  // 16 x v_mov_b64
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      f32x2 v, cc = {c, d};
      asm volatile("v_mov_b64 %0, %1" : "=v"(v) : "v"(cc));
      x[2 * i] += v[0] * 0.f;
    }
  }
}

// mode: 1 = X only, 2 = Y only, 3 = both.  BAR: s_barrier after every iteration.  PRIO: s_setprio 1 on the MFMA waves.
template <int MIX, bool BAR, bool PRIO>
__global__ __launch_bounds__(512, 2) void overlap(unsigned long long* out, const f16x8* src, int iters, int mode) {
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool gx = wave < 4;
  if (!BAR && ((gx && !(mode & 1)) || (!gx && !(mode & 2)))) return;
  const bool work = (gx && (mode & 1)) || (!gx && (mode & 2));
  unsigned long long t0 = 0, t1 = 0;
  if (gx) {
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 7 + i) & 4095]; b[i] = src[(tid * 13 + i + 99) & 4095]; }
    f32x16 o0 = {}, o1 = {}, o2 = {}, o3 = {}, s0 = {}, s1 = {};
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if (work) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], o1, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], o2, 0, 0, 0);
        o3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], o3, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], b[2], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], b[3], o1, 0, 0, 0);
        o2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3], b[2], o2, 0, 0, 0);
        o3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3], b[3], o3, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[3 - k], s1, 0, 0, 0);
        }
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
      }
      if (BAR) __builtin_amdgcn_s_barrier();
    }
    t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += o0[i] + o1[i] + o2[i] + o3[i] + s0[i] + s1[i];
    if ((tid & 63) == 0) out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)s;
  } else {
    float x[32], acc = 0.f;
    unsigned p[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = -0.01f * (float)((tid + i) & 63);
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = 0;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if (work) valu_block<MIX>(x, acc, p, 0.5f, -1.0f);
      if (BAR) __builtin_amdgcn_s_barrier();
    }
    t1 = __builtin_readcyclecounter();
    float s = acc;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (float)p[i];
    if ((tid & 63) == 0) out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)s;
  }
  if ((tid & 63) == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
}

// VALU issue rate of one SIMD as a function of the waves it hosts: every wave of the workgroup runs the VALU block.
template <int MIX>
__global__ void valu_only(unsigned long long* out, int iters) {
  const int tid = threadIdx.x;
  float x[32], acc = 0.f;
  unsigned p[16];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = -0.01f * (float)((tid + i) & 63);
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) valu_block<MIX>(x, acc, p, 0.5f, -1.0f);
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = acc;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += (float)p[i];
  if ((tid & 63) == 0) { out[(blockIdx.x * 16 + (tid >> 6)) * 2] = t1 - t0; out[(blockIdx.x * 16 + (tid >> 6)) * 2 + 1] = (unsigned long long)s; }
}

template <int MIX>
static void run_valu(const char* name, unsigned long long* out, int valu_count) {
  const int iters = 2000, blocks = 256;
  std::vector<unsigned long long> h(blocks * 32);
  printf("%-22s (%3d VALU/iter): cycles per iteration per wave at 1..4 waves per SIMD:", name, valu_count);
  for (int wps = 1; wps <= 4; ++wps) {
    hipMemset(out, 0, blocks * 32 * 8);
    hipLaunchKernelGGL((valu_only<MIX>), dim3(blocks), dim3(256 * wps), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, blocks * 32 * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 4 * wps; ++w) sum += (double)h[(b * 16 + w) * 2] / iters;
    printf("  %6.0f", sum / (blocks * 4 * wps));
  }
  printf("\n");
}

// The loop body of the attention kernel on registers only (no LDS, no global memory, no barrier): 4 dependent score MFMAs,
// the diet softmax on their result, 8 cvt_pk, 4 P.V MFMAs on two accumulators - every wave of the workgroup runs it.
// Tells whether the per-SIMD time of that body is max(MFMA, VALU) (the pipes overlap across waves) or their sum.
__global__ void attn_body(unsigned long long* out, const f16x8* src, int iters) {
  const int tid = threadIdx.x;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f16x8 q[4], k[4], v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { q[i] = src[(tid * 7 + i) & 4095]; k[i] = src[(tid * 13 + i + 99) & 4095]; v[i] = src[(tid * 3 + i + 700) & 4095]; }
  f32x16 o0 = {}, o1 = {};
  float l = 0.f, mc = 1.0f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f32x16 s = {};
#pragma unroll
    for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k[i], q[i], s, 0, 0, 0);
    const f32x2 c2 = {0.18f, 0.18f}, nmc2 = {-mc, -mc};
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x2 sv = {s[2 * i], s[2 * i + 1]};
      const f32x2 e = __builtin_elementwise_fma(sv, c2, nmc2);
      s[2 * i] = __builtin_amdgcn_exp2f(e[0]);
      s[2 * i + 1] = __builtin_amdgcn_exp2f(e[1]);
    }
#pragma unroll
    for (int i = 0; i < 8; i += 2) { a0 += f32x2{s[2 * i], s[2 * i + 1]}; a1 += f32x2{s[2 * i + 2], s[2 * i + 3]}; }
    l += (a0[0] + a0[1]) + (a1[0] + a1[1]);
    f16x8 p0, p1;
#pragma unroll
    for (int e = 0; e < 8; ++e) { p0[e] = (_Float16)s[e]; p1[e] = (_Float16)s[8 + e]; }
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[0], p0, o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[1], p0, o1, 0, 0, 0);
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[2], p1, o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[3], p1, o1, 0, 0, 0);
    asm volatile("" : "+v"(k[0]), "+v"(v[0]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = l;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += o0[i] + o1[i];
  if ((tid & 63) == 0) { out[(blockIdx.x * 16 + (tid >> 6)) * 2] = t1 - t0; out[(blockIdx.x * 16 + (tid >> 6)) * 2 + 1] = (unsigned long long)r; }
}

static void run_body(unsigned long long* out, const f16x8* src) {
  const int iters = 2000, blocks = 256;
  std::vector<unsigned long long> h(blocks * 32);
  printf("attention loop body on registers (8 MFMA = 256 cycles of matrix pipe, ~45 VALU): cycles per iteration per wave / per SIMD at 1..4 waves per SIMD:\n ");
  for (int wps = 1; wps <= 4; ++wps) {
    hipMemset(out, 0, blocks * 32 * 8);
    hipLaunchKernelGGL(attn_body, dim3(blocks), dim3(256 * wps), 0, 0, out, src, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, blocks * 32 * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 4 * wps; ++w) sum += (double)h[(b * 16 + w) * 2] / iters;
    const double per_wave = sum / (blocks * 4 * wps);
    printf("  %d: %6.0f / %6.0f", wps, per_wave, per_wave / wps);
  }
  printf("\n");
}

template <int MIX, bool BAR, bool PRIO>
static void run(const char* name, unsigned long long* out, const f16x8* src, int valu_count) {
  const int iters = 2000, blocks = 256;
  std::vector<unsigned long long> h(blocks * 16);
  double res[4][2] = {};
  for (int mode = 1; mode <= 3; ++mode) {
    hipMemset(out, 0, blocks * 16 * 8);
    hipLaunchKernelGGL((overlap<MIX, BAR, PRIO>), dim3(blocks), dim3(512), 0, 0, out, src, iters, mode);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double sx = 0, sy = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? sx : sy) += (double)h[(b * 8 + w) * 2] / iters;
    res[mode][0] = sx / (blocks * 4);
    res[mode][1] = sy / (blocks * 4);
  }
  printf("%-34s VALU/iter %3d | alone: mfma16 %6.0f cyc, valu %6.0f cyc | together: mfma16 %6.0f, valu %6.0f | sum-alone %6.0f  max-alone %6.0f\n",
         name, valu_count, res[1][0], res[2][1], res[3][0], res[3][1], res[1][0] + res[2][1],
         res[1][0] > res[2][1] ? res[1][0] : res[2][1]);
}

int main() {
  unsigned long long* out; f16x8* src;
  hipMalloc(&out, 256 * 32 * 8); hipMalloc(&src, 4096 * 16);
  std::vector<_Float16> hs(4096 * 8);
  unsigned s = 777u;
  for (auto& v : hs) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 0.001f); }
  hipMemcpy(src, hs.data(), hs.size() * 2, hipMemcpyHostToDevice);
  printf("cycles per iteration (16 MFMA 32x32x16 = 512 cycles of matrix pipe); waves w and w+4 share a SIMD\n");
  run<1, false, false>("fma only", out, src, 32);
  run<2, false, false>("exp only", out, src, 32);
  run<4, false, false>("add chain only", out, src, 32);
  run<8, false, false>("cvt_pk only", out, src, 16);
  run<16, false, false>("max3+permlane only", out, src, 12);
  run<31, false, false>("full softmax mix", out, src, 124);
  run<31, false, true>("full mix, prio 1 on mfma waves", out, src, 124);
  run<31, true, false>("full mix + barrier per iter", out, src, 124);
  run<31, true, true>("full mix + barrier + prio", out, src, 124);
  run<29, false, false>("full mix without exp", out, src, 92);
  run_body(out, src);
  run_valu<1>("fma only", out, 32);
  run_valu<2>("exp only", out, 32);
  run_valu<4>("add chain only", out, 32);
  run_valu<8>("cvt_pk only", out, 16);
  run_valu<31>("full softmax mix", out, 124);
  run_valu<32>("pk_fma x16", out, 16);
  run_valu<64>("pk_add x16 (+3 add)", out, 19);
  run_valu<128>("mov_b64 x16 (+16 fma)", out, 32);
  run_valu<2 + 8 + 32 + 64>("diet mix: pkfma16 exp32 pkadd16 cvt16", out, 83);
  return 0;
}
