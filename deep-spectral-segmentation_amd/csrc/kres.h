// kres.h - pieces of the K-resident Linear kernel (linear384.hip): LDS-DMA helper types and the
// exact-erf GELU on packed fp32.
#pragma once
#include "common.h"

namespace dss {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// GELU(x) = x * 0.5 * (1 + erf(x / sqrt 2)), NP float2 at a time, in place.  erf by A&S 7.1.28 on
// z = |x| / sqrt 2: erf z = 1 - q^-16, q = 1 + a1 z + ... + a6 z^6 (|error| <= 3e-7: one v_rcp, no v_exp).  Evaluated
// as x/2 + (|x|/2) (1 - q^-16): the negative branch cancels to -(|x|/2) q^-16 with absolute error ~6e-8 |x|, far
// below the f16 rounding of the output.  The chain of one value is 17 dependent VALU ops (the probe measures
// latency-, not issue-bound execution), so NP pairs are advanced in lockstep: every step below is NP independent
// v_pk_* instructions.
template <int NP>
__device__ __forceinline__ void gelu_erf2xn(f32x2* x) {
  f32x2 z[NP], q[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    f32x2 ax;
    ax[0] = fabsf(x[j][0]);
    ax[1] = fabsf(x[j][1]);
    z[j] = ax * 0.70710678118654752f;
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = z[j] * 0.0000430638f + 0.0002765672f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 0.0001520143f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 0.0092705272f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 0.0422820123f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 0.0705230784f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 1.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {                            // q^16 (inf for |x| > ~30: 1/inf = 0, erf = 1)
#pragma unroll
    for (int j = 0; j < NP; ++j) q[j] = q[j] * q[j];
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    q[j][0] = __builtin_amdgcn_rcpf(q[j][0]);
    q[j][1] = __builtin_amdgcn_rcpf(q[j][1]);
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = (1.0f - q[j]) * (z[j] * 0.70710678118654752f);   // (|x|/2) erf
#pragma unroll
  for (int j = 0; j < NP; ++j) x[j] = x[j] * 0.5f + q[j];
}

// GELU on PACKED f16 (round 5; `gelu = 2` of the Linear entry points, f16 operands only): NP registers of two values each, in
// place.  Why: the fp32 form above is 17.5 one-result instructions per value, a wave issues one VALU instruction per ~5 cycles
// whatever its partner does (profiles/r04_linear_lab.txt item 8), and fc1's epilogue was as long as its MFMA phase; packed fp32
// does not co-issue with the other wave's MFMAs at all.  v_pk_*_f16 does, and this form is 5.5 instructions per value:
//     a = min(|x|, 4.25);  t = a / 4.25;  q = c0 + c1 t + ... + c6 t^6 (Horner, six v_pk_fma_f16);  Phi(-a) ~ q^2;
//     GELU(x) = max(x, 0) - a Phi(-a) = fma(-a, q q, max(x, 0))
// (the even part of GELU is relu(x); what is left, -a Phi(-a), lies in [-0.17, 0] and is the only thing approximated: for
// x > 0 its error is absorbed by the rounding of the final sum, for x -> 0 it is relative to a; beyond |x| = 4.25 the term is held
// at its value there, -1.5e-4 as evaluated, where the exact one decays to 0).  q = sqrt(Phi(-a)) is fitted,
// not Phi itself: the square keeps the far tail non-negative and relatively accurate.  Coefficients: least squares on
// [0, 4.25] with c0 pinned to f16(sqrt(1/2)), rounded to f16 (tests/util.gelu_f16_poly is the same arithmetic in numpy).
// PARITY COST, measured over every f16 input (tests/test_host_logic.py::test_gelu_f16_poly_error_budget): max |error| 1.1e-3
// (at x = 2.6, 0.57 of the output's f16 spacing there; in units of that spacing the worst case is 2.1 for 0.25 < x < 0.5, 1.6 up to
// 1, 1.1 up to 2 - a correctly rounded value is within 0.5), <= 3.2e-4 for x < 0, relative error <= 2e-3 for |x| < 0.5; 54 % of the
// outputs are the correctly rounded f16 value; noise-to-signal for Gaussian pre-activations of sigma 0.7 / 1.5 / 3:
// 2.9e-4 / 2.2e-4 / 1.3e-4 against 2.0e-4 / 2.0e-4 / 1.2e-4 for the correctly rounded f16 output the fp32 form delivers.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 h2_bits(unsigned short b) {
  const _Float16 v = __builtin_bit_cast(_Float16, b);
  return h2{v, v};
}
template <int NP>
__device__ __forceinline__ void gelu_poly_f16xn(h2* x) {
  h2 a[NP], t[NP], q[NP], m[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) a[j] = __builtin_elementwise_min(__builtin_elementwise_abs(x[j]), h2_bits(0x4440));   // min(|x|, 4.25)
#pragma unroll
  for (int j = 0; j < NP; ++j) t[j] = a[j] * h2_bits(0x3388);                                                    // a / 4.25 in [0, 1]
#pragma unroll
  for (int j = 0; j < NP; ++j) asm("v_pk_max_f16 %0, %1, 0" : "=v"(m[j]) : "v"(x[j]));   // (asm: the builtin canonicalises x first)
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = __builtin_elementwise_fma(h2_bits(0x3f09), t[j], h2_bits(0xc3b6));
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = __builtin_elementwise_fma(q[j], t[j], h2_bits(0x3c3f));
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = __builtin_elementwise_fma(q[j], t[j], h2_bits(0x4167));
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = __builtin_elementwise_fma(q[j], t[j], h2_bits(0xbcb7));
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = __builtin_elementwise_fma(q[j], t[j], h2_bits(0xbcc0));
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = __builtin_elementwise_fma(q[j], t[j], h2_bits(0x39a8));
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * q[j];
#pragma unroll
  for (int j = 0; j < NP; ++j) x[j] = __builtin_elementwise_fma(-a[j], q[j], m[j]);
}

}  // namespace dss
