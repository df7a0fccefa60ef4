// TEST INFRASTRUCTURE ONLY - single-thread host emulation of csrc/eigs_core.h (the block-cooperative
// Lanczos the GPU kernel runs), built by tests/ with g++ so the restart / Rayleigh-Ritz logic can be
// checked against tests/golden on a machine without a GPU.  Not part of libdss_hip.so; the product
// never loads this.
#define DSS_HOST_EMUL 1
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../deep-spectral-segmentation_amd/csrc/eigs_core.h"

// W: packed upper-triangular storage (csrc/eigs_core.h wsym_layout), wsym_elems(N) elements per image; WE = float
// (the values) or uint16_t (round(65535 w), the default storage of the product path).
template <class WE>
static int emul(const WE* W, int B, int N, int ld, int K, float* eigenvalues, float* eigenvectors, int32_t* info,
                int ncv, int keep, float tol, int max_restarts, int mode) {
  using namespace dss;
  if (ncv > EIGS_MAX_NCV) ncv = EIGS_MAX_NCV;
  if (ncv > N) ncv = N;
  EigsParams P;
  P.N = N; P.ld = ld; P.K = K; P.ncv = ncv; P.keep = keep; P.max_restarts = max_restarts; P.tol = tol; P.mode = mode;
  const EigsLds L = eigs_lds_layout(ld, ncv);
  std::vector<unsigned char> lds(L.total + 64);
  unsigned char* lp = lds.data();
  lp += (16 - ((uintptr_t)lp & 15)) & 15;
  std::vector<float> gws(eigs_ws_floats_per_image(ld, ncv));
  for (int b = 0; b < B; ++b) {
    memset(lp, 0, L.total);
    eigs_one_image(W + (size_t)b * wsym_elems(N), P, gws.data(), lp, eigenvalues + (size_t)b * K,
                   eigenvectors + (size_t)b * K * N, info + b);
  }
  return 0;
}

extern "C" int dss_emul_laplacian_eigs(const float* W, int B, int N, int ld, int K, float* eigenvalues,
                                       float* eigenvectors, int32_t* info, int ncv, int keep, float tol,
                                       int max_restarts, int mode) {
  return emul<float>(W, B, N, ld, K, eigenvalues, eigenvectors, info, ncv, keep, tol, max_restarts, mode);
}

extern "C" int dss_emul_laplacian_eigs_u16(const uint16_t* W, int B, int N, int ld, int K, float* eigenvalues,
                                           float* eigenvectors, int32_t* info, int ncv, int keep, float tol,
                                           int max_restarts) {
  return emul<uint16_t>(W, B, N, ld, K, eigenvalues, eigenvectors, info, ncv, keep, tol, max_restarts, 0);
}

// the storage layout itself, for the tests' independent restatement of it (tests/util.pack_sym)
extern "C" long dss_emul_wsym_elems(int N) { return (long)dss::wsym_elems(N); }
extern "C" long dss_emul_wsym_at(int N, int ti, int tj, int lr, int lc) {
  const dss::WsymLayout L = dss::wsym_layout(N);
  return dss::wsym_has(L, tj, lc) ? (long)dss::wsym_at(L, ti, tj, lr, lc) : -1;
}
