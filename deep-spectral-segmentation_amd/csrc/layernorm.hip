// layernorm.hip - row LayerNorm for the ViT residual stream (optionally fused with the residual add).
//
// Replaces torch.nn.LayerNorm(D, eps=1e-6) as DINO's Block uses it (norm1/norm2; SURVEY.md
// Appendix A; reached from extract/extract.py:94) and the `x = x + f(...)` add that precedes it.
//
// HBM-bound: per element 4 B read (+2 B residual) and 2 B written (+4 B when the residual stream is
// updated in place).  One 64-lane wave owns one row: the row lives in registers as float4s
// (16 B/lane coalesced loads), mean and the centred variance are two wave butterflies (no LDS, no
// barrier), and the output is written as packed halves (8 B/lane).
#include "common.h"

namespace dss {

template <class T> struct pack4 {
  __device__ static void store(T* p, f32x4 v) {
    typename vec4<T>::type o;
    o[0] = from_f32<T>(v[0]); o[1] = from_f32<T>(v[1]); o[2] = from_f32<T>(v[2]); o[3] = from_f32<T>(v[3]);
    *reinterpret_cast<typename vec4<T>::type*>(p) = o;
  }
  __device__ static f32x4 load(const T* p) {
    typename vec4<T>::type i = *reinterpret_cast<const typename vec4<T>::type*>(p);
    f32x4 v = {to_f32<T>(i[0]), to_f32<T>(i[1]), to_f32<T>(i[2]), to_f32<T>(i[3])};
    return v;
  }
};

// VPL = float4 vectors per lane (row length D <= VPL * 256).  RES: residual dtype (void = none).
template <int VPL, class TO, class TR, bool HAS_RES>
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, const TR* __restrict__ res,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        TO* __restrict__ y, int rows, int D, float eps,
                                                        long res_ld, long res_plane) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nvec = D >> 2;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    float* xr = x + row * D;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        if (HAS_RES) {
          // vector c = columns 4c .. 4c+3 = 64-column group c >> 4, offset 4 (c & 15).  Row-major residual:
          // (res_ld, res_plane) = (D, 64); planar [D/64][rows][64] (dss_linear_k384's DSS_PLANAR64): (64, 64 rows)
          v[i] += pack4<TR>::load(res + (c >> 4) * res_plane + row * res_ld + 4 * (c & 15));
          *reinterpret_cast<f32x4*>(xr + 4 * c) = v[i];
        }
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      } else {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        f32x4 d = v[i] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * c);
        f32x4 o = (v[i] - mean) * rstd * g + b;
        pack4<TO>::store(y + row * D + 4 * c, o);
      }
    }
  }
}

template <int VPL, class TO, class TR, bool HAS_RES>
static void launch_ln(float* x, const void* res, const float* gamma, const float* beta, void* y, int rows,
                      int D, float eps, int planar, hipStream_t s) {
  int blocks = ceil_div(rows, 4);
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL((layernorm_kernel<VPL, TO, TR, HAS_RES>), dim3(blocks), dim3(256), 0, s, x,
                     (const TR*)res, gamma, beta, (TO*)y, rows, D, eps, planar ? 64L : (long)D,
                     planar ? 64L * rows : 64L);
}

template <class TO, class TR, bool HAS_RES>
static void dispatch_vpl(float* x, const void* res, const float* gamma, const float* beta, void* y, int rows,
                         int D, float eps, int planar, hipStream_t s) {
  const int vpl = ceil_div(D, 256);
  switch (vpl) {
    case 1: launch_ln<1, TO, TR, HAS_RES>(x, res, gamma, beta, y, rows, D, eps, planar, s); break;
    case 2: launch_ln<2, TO, TR, HAS_RES>(x, res, gamma, beta, y, rows, D, eps, planar, s); break;
    case 3: launch_ln<3, TO, TR, HAS_RES>(x, res, gamma, beta, y, rows, D, eps, planar, s); break;
    case 4: launch_ln<4, TO, TR, HAS_RES>(x, res, gamma, beta, y, rows, D, eps, planar, s); break;
    default: launch_ln<8, TO, TR, HAS_RES>(x, res, gamma, beta, y, rows, D, eps, planar, s); break;
  }
}

template <class TO>
static int dispatch_res(float* x, const void* res, int res_dtype, const float* gamma, const float* beta,
                        void* y, int rows, int D, float eps, int planar, hipStream_t s) {
  if (!res) {
    dispatch_vpl<TO, float, false>(x, nullptr, gamma, beta, y, rows, D, eps, 0, s);
    return DSS_OK;
  }
  switch (res_dtype) {
    case DSS_F32: dispatch_vpl<TO, float, true>(x, res, gamma, beta, y, rows, D, eps, planar, s); return DSS_OK;
    case DSS_F16: dispatch_vpl<TO, f16, true>(x, res, gamma, beta, y, rows, D, eps, planar, s); return DSS_OK;
    case DSS_BF16: dispatch_vpl<TO, bf16, true>(x, res, gamma, beta, y, rows, D, eps, planar, s); return DSS_OK;
  }
  return fail(DSS_ERR_BAD_ARG, "dss_layernorm_fwd: unsupported res_dtype %d", res_dtype);
}

}  // namespace dss

extern "C" int dss_layernorm_fwd(float* x, const void* residual, int res_dtype, int res_layout, const float* gamma,
                                 const float* beta, void* y, int out_dtype, int rows, int D, float eps,
                                 void* stream) {
  DSS_REQUIRE(x && gamma && beta && y, "dss_layernorm_fwd: null pointer");
  DSS_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 2048,
              "dss_layernorm_fwd: need rows > 0, D %% 4 == 0, D <= 2048 (rows=%d D=%d)", rows, D);
  DSS_REQUIRE(res_layout == DSS_ROW_MAJOR || (res_layout == DSS_PLANAR64 && D % 64 == 0),
              "dss_layernorm_fwd: res_layout must be DSS_ROW_MAJOR, or DSS_PLANAR64 with D %% 64 == 0 (got %d, D=%d)",
              res_layout, D);
  const int planar = residual && res_layout == DSS_PLANAR64;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  switch (out_dtype) {
    case DSS_F32: rc = dss::dispatch_res<float>(x, residual, res_dtype, gamma, beta, y, rows, D, eps, planar, s); break;
    case DSS_F16: rc = dss::dispatch_res<dss::f16>(x, residual, res_dtype, gamma, beta, y, rows, D, eps, planar, s); break;
    case DSS_BF16: rc = dss::dispatch_res<dss::bf16>(x, residual, res_dtype, gamma, beta, y, rows, D, eps, planar, s); break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_layernorm_fwd: unsupported out_dtype %d", out_dtype);
  }
  if (rc != DSS_OK) return rc;
  DSS_CHECK_LAUNCH("layernorm");
  return DSS_OK;
}
