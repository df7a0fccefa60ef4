// kres_r4_lab.h - lab snapshot of csrc/kres.h with the DSS_GELU_SCALAR branch (see linear384_r4_lab.hip).
// kres.h - pieces of the K-resident Linear kernel (linear384.hip): LDS-DMA helper types and the
// exact-erf GELU on packed fp32.
#pragma once
#include "../../deep-spectral-segmentation_amd/csrc/common.h"

namespace dss {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// GELU(x) = x * 0.5 * (1 + erf(x / sqrt 2)), NP float2 at a time, in place.  erf by A&S 7.1.28 on
// z = |x| / sqrt 2: erf z = 1 - q^-16, q = 1 + a1 z + ... + a6 z^6 (|error| <= 3e-7: one v_rcp, no v_exp).  Evaluated
// as x/2 + (|x|/2) (1 - q^-16): the negative branch cancels to -(|x|/2) q^-16 with absolute error ~6e-8 |x|, far
// below the f16 rounding of the output.  The chain of one value is 17 dependent VALU ops (the probe measures
// latency-, not issue-bound execution), so NP pairs are advanced in lockstep: every step below is NP independent
// v_pk_* instructions.
#ifdef DSS_GELU_SCALAR   // lab (build with -fno-slp-vectorize): the same arithmetic on one-result instructions
template <int NP>
__device__ __forceinline__ void gelu_erf2xn(f32x2* x) {
  float z[2 * NP], q[2 * NP];
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) z[j] = fabsf(x[j >> 1][j & 1]) * 0.70710678118654752f;
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) q[j] = fmaf(z[j], 0.0000430638f, 0.0002765672f);
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) q[j] = fmaf(q[j], z[j], 0.0001520143f);
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) q[j] = fmaf(q[j], z[j], 0.0092705272f);
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) q[j] = fmaf(q[j], z[j], 0.0422820123f);
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) q[j] = fmaf(q[j], z[j], 0.0705230784f);
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) q[j] = fmaf(q[j], z[j], 1.0f);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j) q[j] = q[j] * q[j];
  }
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) q[j] = __builtin_amdgcn_rcpf(q[j]);
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) q[j] = (1.0f - q[j]) * (z[j] * 0.70710678118654752f);
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) x[j >> 1][j & 1] = fmaf(x[j >> 1][j & 1], 0.5f, q[j]);
}
#else
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
#endif

}  // namespace dss
