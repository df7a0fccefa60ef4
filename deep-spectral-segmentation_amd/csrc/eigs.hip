// eigs.hip - kernel wrappers + C ABI for the Laplacian eigen stage (algorithm: eigs_core.h) and the
// stand-alone sign rule.
#include "common.h"
#include "eigs_core.h"

namespace dss {

static constexpr int EIGS_THREADS = 512;          // launch size: 8 waves per image
static constexpr int EIGS_WAVES_PER_SIMD = 4;    // 128 registers per lane: two images per CU

template <class WE>
__global__ __launch_bounds__(EIGS_THREADS, EIGS_WAVES_PER_SIMD) void laplacian_eigs_kernel(const WE* __restrict__ W, EigsParams P,
                                                                      float* gws, size_t gws_stride,
                                                                      float* eigenvalues, float* eigenvectors,
                                                                      int32_t* info) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const size_t b = blockIdx.x;
  eigs_one_image(W + b * wsym_elems(P.N), P, gws + b * gws_stride, lds, eigenvalues + b * P.K,
                 eigenvectors + b * (size_t)P.K * P.N, info + b);
}

// extract/extract.py:238-240 on its own: one workgroup per vector.
__global__ __launch_bounds__(256) void sign_rule_kernel(float* v, int N) {
  __shared__ int cnt;
  float* row = v + (size_t)blockIdx.x * N;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int pos = 0;
  for (int e = threadIdx.x; e < N; e += blockDim.x) pos += row[e] > 0.f ? 1 : 0;
  pos = (int)(wave_sum((float)pos) + 0.5f);
  if ((threadIdx.x & 63) == 0) atomicAdd(&cnt, pos);
  __syncthreads();
  const int c = cnt;
  if (2 * c > N && c < N)
    for (int e = threadIdx.x; e < N; e += blockDim.x) row[e] = 0.f - row[e];
}

static int resolve_ncv(int N, int K, int ncv) {
  if (ncv <= 0) ncv = 2 * K + 10 > 20 ? 2 * K + 10 : 20;
  if (ncv > EIGS_MAX_NCV) ncv = EIGS_MAX_NCV;
  if (ncv > N) ncv = N;
  return ncv;
}

}  // namespace dss

extern "C" size_t dss_eigs_workspace_bytes(int B, int N, int K, int ncv) {
  if (B <= 0 || N <= 0 || K <= 0) return 0;
  ncv = dss::resolve_ncv(N, K, ncv);
  return (size_t)B * dss::eigs_ws_floats_per_image(dss_affinity_ld(N), ncv) * sizeof(float);
}

namespace dss {

template <class WE>
static int symmetric_eigs(const WE* W, int B, int N, int K, int mode, float* eigenvalues, float* eigenvectors,
                          int32_t* info, int ncv, float tol, int max_restarts, void* workspace, size_t workspace_bytes,
                          void* stream) {
  DSS_REQUIRE(W && eigenvalues && eigenvectors && info && workspace, "dss_laplacian_eigs: null pointer");
  DSS_REQUIRE(mode == DSS_EIGS_NORMALIZED_LAPLACIAN || mode == DSS_EIGS_AFFINITY_LM || mode == DSS_EIGS_LAPLACIAN,
              "dss_symmetric_eigs: unknown mode %d", mode);
  DSS_REQUIRE(B > 0 && N > 1 && K > 0, "dss_laplacian_eigs: bad shape B=%d N=%d K=%d", B, N, K);
  DSS_REQUIRE(K < N, "dss_laplacian_eigs: need K < N (K=%d, N=%d)", K, N);
  ncv = resolve_ncv(N, K, ncv);
  DSS_REQUIRE(ncv >= K + 2 || ncv == N,
              "dss_laplacian_eigs: Krylov dimension %d too small for K=%d (max %d)", ncv, K, EIGS_MAX_NCV);
  const int ld = dss_affinity_ld(N);
  const size_t per_img = eigs_ws_floats_per_image(ld, ncv);
  if (workspace_bytes < (size_t)B * per_img * sizeof(float))
    return fail(DSS_ERR_WORKSPACE, "dss_laplacian_eigs: workspace %zu < %zu bytes", workspace_bytes,
                (size_t)B * per_img * sizeof(float));
  EigsParams P;
  P.N = N; P.ld = ld; P.K = K; P.ncv = ncv;
  P.keep = (ncv + K) / 2;  // tuned on tests/golden with the host emulation (tests/host_emul)
  P.max_restarts = max_restarts > 0 ? max_restarts : 60;
  P.tol = tol > 0.f ? tol : 2e-6f;
  P.mode = mode;
  const EigsLds L = eigs_lds_layout(ld, ncv);
  DSS_REQUIRE(L.total <= 160 * 1024, "dss_laplacian_eigs: N=%d needs %zu B of LDS (> 160 KiB)", N, L.total);
  hipError_t e = hipFuncSetAttribute((const void*)laplacian_eigs_kernel<WE>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
  if (e != hipSuccess)
    return fail(DSS_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS=%zu): %s", L.total, hipGetErrorString(e));
  // 512-thread workgroups let two images share a CU: the serial Rayleigh-Ritz / restart phases of one overlap the W
  // streaming of the other (measured against one 16-wave workgroup per CU in round 1).
#ifdef DSS_EIGS_THREADS   // lab builds (scripts/debug/eigs_lab.py)
  const int threads = DSS_EIGS_THREADS;
#else
  const int threads = 512;
#endif
  hipLaunchKernelGGL(laplacian_eigs_kernel<WE>, dim3(B), dim3(threads), L.total, (hipStream_t)stream, W, P,
                     (float*)workspace, per_img, eigenvalues, eigenvectors, info);
  DSS_CHECK_LAUNCH("laplacian_eigs");
  return DSS_OK;
}

}  // namespace dss

extern "C" int dss_laplacian_eigs(const float* W, int B, int N, int K, float* eigenvalues, float* eigenvectors,
                                  int32_t* info, int ncv, float tol, int max_restarts, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return dss::symmetric_eigs<float>(W, B, N, K, DSS_EIGS_NORMALIZED_LAPLACIAN, eigenvalues, eigenvectors, info, ncv,
                                    tol, max_restarts, workspace, workspace_bytes, stream);
}

extern "C" int dss_laplacian_eigs_u16(const uint16_t* W, int B, int N, int K, float* eigenvalues,
                                      float* eigenvectors, int32_t* info, int ncv, float tol, int max_restarts,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return dss::symmetric_eigs<uint16_t>(W, B, N, K, DSS_EIGS_NORMALIZED_LAPLACIAN, eigenvalues, eigenvectors, info,
                                       ncv, tol, max_restarts, workspace, workspace_bytes, stream);
}

extern "C" int dss_symmetric_eigs(const float* W, int B, int N, int K, int mode, float* eigenvalues,
                                  float* eigenvectors, int32_t* info, int ncv, float tol, int max_restarts,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  return dss::symmetric_eigs<float>(W, B, N, K, mode, eigenvalues, eigenvectors, info, ncv, tol, max_restarts,
                                    workspace, workspace_bytes, stream);
}

#ifdef DSS_EIGS_TIMELINE
extern "C" int dss_eigs_timeline(unsigned long long* out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(dss_eigs_tl), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(dss_eigs_tl), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

#ifdef DSS_EIGS_RHO_DEBUG
extern "C" int dss_eigs_rho_buffer(float* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(dss_eigs_rho_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int dss_sign_rule(float* eigenvectors, int rows, int N, void* stream) {
  DSS_REQUIRE(eigenvectors, "dss_sign_rule: null pointer");
  DSS_REQUIRE(rows > 0 && N > 0, "dss_sign_rule: bad shape rows=%d N=%d", rows, N);
  hipLaunchKernelGGL(dss::sign_rule_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, eigenvectors, N);
  DSS_CHECK_LAUNCH("sign_rule");
  return DSS_OK;
}
