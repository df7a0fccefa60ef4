// lt_tune_probe.hip - round 6: is hipBLASLt's first heuristic candidate the fastest kernel it has for the library-GEMM shapes?
//
// dss_linear_lt (csrc/gemm.hip) takes the first candidate of hipblasLtMatmulAlgoGetHeuristic that runs without a partial-tile
// workspace.  This probe poses the same problem (C[M, N] = A[M, K] W[N, K]^T + bias, f16, Tensile's data-parallel switch set),
// enumerates EVERY solution of the type combination (hipblaslt_ext::getAllAlgos), keeps those that support the problem with a
// workspace of 0 bytes, times each (1 warm-up + 3 launches, HIP events, min) and prints them fastest first beside the
// heuristic's own order.
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/lt_tune_probe.hip -o scripts/probes/lt_tune_probe -lhipblaslt
//     scripts/probes/lt_tune_probe M N K [out: 0 = f16 | 1 = f32 accumulate (beta = 1, fp32 bias)] [max seconds] [M2 M3 ...]
// (further M values: the first M's twelve fastest and the heuristic's first four of THAT M are timed there as well.)  Operands are
// random; every timed solution's output is check-summed and compared with the heuristic's first choice ('=' same bits, '~' not).
// The library that answers is the one the loader finds as libhipblaslt.so.1: the stand-alone program gets /opt/rocm's, the product
// inside a Python process binds to the build PyTorch bundles (torch/lib/libhipblaslt.so, loaded first, same soname) - a DIFFERENT
// set of solutions.  To ask that one: build with -shared -fPIC -DLT_TUNE_AS_LIBRARY and run scripts/debug/lt_tune.py.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

__global__ void fill_f16(unsigned short* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const _Float16 v = (_Float16)(((int)(h & 2047) - 1024) * (1.0f / 4096.0f));      // multiples of 2^-12 in [-0.25, 0.25)
    p[i] = *reinterpret_cast<const unsigned short*>(&v);
  }
}
__global__ void checksum_u32(const unsigned* p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += (unsigned long long)p[i] * (unsigned)(2 * (i % 1021) + 1);
  atomicAdd(out, acc);
}

#define CK(x) do { auto s__ = (x); if ((int)s__ != 0) { printf("%s -> %d\n", #x, (int)s__); return 1; } } while (0)

static std::string tile_of(const std::string& name) {
  size_t p = name.find("_MT");
  if (p == std::string::npos) return "?";
  size_t q = name.find('_', p + 1);
  std::string t = name.substr(p + 1, q - p - 1);
  for (const char* key : {"_MI", "_LDSB", "_GSU", "_WG", "_WGM", "_PGR", "_PLR", "_DTL", "_1LDSB", "_SU", "_SUM", "_WS", "_NTA", "_NTB", "_NTC", "_NTD", "_SVW", "_TLDS"}) {
    size_t a = name.find(key);
    while (a != std::string::npos) {
      size_t b = name.find('_', a + 1);
      std::string f = name.substr(a + 1, (b == std::string::npos ? name.size() : b) - a - 1);
      if (f.size() > strlen(key) - 1 && (isdigit((unsigned char)f[strlen(key) - 1]) || f[strlen(key) - 1] == 'x')) { t += " " + f; break; }
      a = name.find(key, a + 1);
    }
  }
  return t;
}

#ifdef LT_TUNE_AS_LIBRARY
extern "C" int lt_tune_main(int argc, char** argv) {      // called from scripts/debug/lt_tune.py inside a process that imported torch
#else
int main(int argc, char** argv) {
#endif
  if (argc < 4) { printf("usage: lt_tune_probe M N K [out 0|1] [max seconds]\n"); return 2; }
  const long M = atol(argv[1]);
  const int N = atoi(argv[2]), K = atoi(argv[3]);
  const int acc = argc > 4 ? atoi(argv[4]) : 0;
  const double max_s = argc > 5 ? atof(argv[5]) : 60.0;
  setenv("TENSILE_STREAMK_DATA_PARALLEL", "1", 0);
  hipblasLtHandle_t h;
  CK(hipblasLtCreate(&h));
  void *A, *W, *C, *bias;
  CK(hipMalloc(&A, (size_t)M * K * 2));
  CK(hipMalloc(&W, (size_t)N * K * 2));
  CK(hipMalloc(&C, (size_t)M * N * (acc ? 4 : 2)));
  CK(hipMalloc(&bias, (size_t)N * 4));
  hipLaunchKernelGGL(fill_f16, dim3(4096), dim3(256), 0, 0, (unsigned short*)A, (size_t)M * K, 1u);
  hipLaunchKernelGGL(fill_f16, dim3(1024), dim3(256), 0, 0, (unsigned short*)W, (size_t)N * K, 2u);
  unsigned long long* dsum;
  CK(hipMalloc(&dsum, 8));
  CK(hipMemset(C, 0, (size_t)M * N * (acc ? 4 : 2)));
  CK(hipMemset(bias, 0, (size_t)N * 4));

  hipblasLtMatmulDesc_t desc;
  CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
  const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
  const hipDataType bt = acc ? HIP_R_32F : HIP_R_16F, ct = acc ? HIP_R_32F : HIP_R_16F;
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
  hipblasLtMatrixLayout_t la, lb, lc;
  CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, (uint64_t)K, (uint64_t)N, (int64_t)K));
  CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, (uint64_t)K, (uint64_t)M, (int64_t)K));
  CK(hipblasLtMatrixLayoutCreate(&lc, ct, (uint64_t)N, (uint64_t)M, (int64_t)N));
  const float alpha = 1.f, beta = acc ? 1.f : 0.f;

  // the heuristic's own order (what dss_linear_lt walks)
  hipblasLtMatmulPreference_t pref;
  CK(hipblasLtMatmulPreferenceCreate(&pref));
  size_t budget = (size_t)128 << 20;
  CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &budget, sizeof(budget)));
  std::vector<hipblasLtMatmulHeuristicResult_t> heur(32);
  int got = 0;
  CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 32, heur.data(), &got));
  std::vector<int> heur_idx;
  for (int i = 0; i < got; ++i) heur_idx.push_back(hipblaslt_ext::getIndexFromAlgo(heur[i].algo));

  std::vector<hipblasLtMatmulHeuristicResult_t> all;
  CK(hipblaslt_ext::getAllAlgos(h, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, ta, tb, HIP_R_16F, HIP_R_16F, ct, ct, HIPBLAS_COMPUTE_32F, all));
  printf("=== M %ld N %d K %d %s: %zu solutions of the type combination, heuristic returned %d\n", M, N, K, acc ? "fp32 accumulate (beta = 1)" : "f16 out",
         all.size(), got);

  struct Row { double us; int idx, rank; unsigned long long sum; std::string name; };
  std::vector<Row> rows;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t c_words = (size_t)M * N * (acc ? 4 : 2) / 4;

  // min of three launches behind one warm-up; `sum` (optional): the checksum of the output of one more launch into a zeroed C
  auto time_algo = [&](hipblasLtMatmulAlgo_t& algo, hipblasLtMatrixLayout_t lbm, hipblasLtMatrixLayout_t lcm, double give_up_us, unsigned long long* sum) -> double {
    double best = 1e30;
    for (int it = 0; it < 4; ++it) {
      hipEventRecord(e0, 0);
      if (hipblasLtMatmul(h, desc, &alpha, W, la, A, lbm, &beta, C, lcm, C, lcm, &algo, nullptr, 0, 0) != HIPBLAS_STATUS_SUCCESS) return -1;
      hipEventRecord(e1, 0);
      if (hipEventSynchronize(e1) != hipSuccess) return -2;
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (it > 0 && ms * 1e3 < best) best = ms * 1e3;
      if (it == 0 && ms * 1e3 > give_up_us) return ms * 1e3;           // hopeless: one launch is enough
    }
    if (sum) {
      hipMemsetAsync(C, 0, c_words * 4, 0);
      hipMemsetAsync(dsum, 0, 8, 0);
      if (hipblasLtMatmul(h, desc, &alpha, W, la, A, lbm, &beta, C, lcm, C, lcm, &algo, nullptr, 0, 0) != HIPBLAS_STATUS_SUCCESS) return -1;
      hipLaunchKernelGGL(checksum_u32, dim3(2048), dim3(256), 0, 0, (const unsigned*)C, c_words, dsum);
      if (hipMemcpy(sum, dsum, 8, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    }
    return best;
  };

  int supported = 0, with_ws = 0;
  const auto t_start = std::chrono::steady_clock::now();
  // the heuristic's candidates first (they must be in the table whatever the time limit), then everything else
  std::vector<hipblasLtMatmulHeuristicResult_t> order(heur.begin(), heur.begin() + got);
  for (auto& r : all) {
    const int idx = hipblaslt_ext::getIndexFromAlgo(r.algo);
    if (std::find(heur_idx.begin(), heur_idx.end(), idx) == heur_idx.end()) order.push_back(r);
  }
  size_t visited = 0;
  double fastest = 1e30;
  for (auto& r : order) {
    ++visited;
    size_t ws = 0;
    if (hipblaslt_ext::matmulIsAlgoSupported(h, desc, &alpha, la, lb, &beta, lc, lc, r.algo, ws) != HIPBLAS_STATUS_SUCCESS) continue;
    ++supported;
    if (ws != 0) { ++with_ws; continue; }
    const int idx = hipblaslt_ext::getIndexFromAlgo(r.algo);
    unsigned long long sum = 0;
    const double us = time_algo(r.algo, lb, lc, (int)visited > got ? 3 * fastest : 1e30, &sum);
    if (us == -2) { printf("launch failed for idx %d\n", idx); return 1; }
    if (us < 0) continue;
    fastest = std::min(fastest, us);
    const auto pos = std::find(heur_idx.begin(), heur_idx.end(), idx);
    rows.push_back({us, idx, pos == heur_idx.end() ? -1 : (int)(pos - heur_idx.begin()), sum, hipblaslt_ext::getSolutionNameFromAlgo(h, r.algo)});
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > max_s && (int)visited >= got) break;
  }
  std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.us < b.us; });
  printf("    visited %zu of %zu, %d support the problem, %d of those ask for a workspace (skipped), %zu timed\n", visited, order.size(), supported, with_ws,
         rows.size());
  const double flops = 2.0 * M * N * K;
  double first = 0;
  unsigned long long first_sum = 0;
  for (auto& r : rows) if (r.rank == 0) { first = r.us; first_sum = r.sum; }
  size_t same = 0;
  for (auto& r : rows) same += r.sum == first_sum;
  printf("    %zu of %zu timed solutions return the bits of the heuristic's first choice\n", same, rows.size());
  for (size_t i = 0; i < rows.size() && i < 12; ++i)
    printf("    %9.1f us  %6.1f TFLOP/s  %+6.1f %% vs heuristic #0  %c  idx %d  heuristic rank %2d   %s\n", rows[i].us, flops / rows[i].us / 1e6,
           first > 0 ? 100.0 * (rows[i].us / first - 1.0) : 0.0, rows[i].sum == first_sum ? '=' : '~', rows[i].idx, rows[i].rank, tile_of(rows[i].name).c_str());
  for (auto& r : rows)
    if (r.rank >= 0 && r.rank < 4)
      printf("    heuristic #%d: %9.1f us  %c  idx %d  %s\n", r.rank, r.us, r.sum == first_sum ? '=' : '~', r.idx, tile_of(r.name).c_str());
  for (size_t i = 0; i < rows.size() && i < 3; ++i) printf("    full name of #%zu (idx %d): %s\n", i + 1, rows[i].idx, rows[i].name.c_str());

  // the same solutions at other row counts
  for (int a = 6; a < argc; ++a) {
    const long M2 = atol(argv[a]);
    if (M2 <= 0 || M2 > M) { printf("    (M2 = %ld skipped: must be in 1..M)\n", M2); continue; }
    hipblasLtMatrixLayout_t lb2, lc2;
    CK(hipblasLtMatrixLayoutCreate(&lb2, HIP_R_16F, (uint64_t)K, (uint64_t)M2, (int64_t)K));
    CK(hipblasLtMatrixLayoutCreate(&lc2, ct, (uint64_t)N, (uint64_t)M2, (int64_t)N));
    std::vector<hipblasLtMatmulHeuristicResult_t> heur2(8);
    int got2 = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb2, lc2, lc2, pref, 8, heur2.data(), &got2));
    printf("  --- the same at M = %ld\n", M2);
    double first2 = 0;
    for (int i = 0; i < got2 && i < 4; ++i) {
      const double us = time_algo(heur2[i].algo, lb2, lc2, 1e30, nullptr);
      if (i == 0) first2 = us;
      printf("    heuristic #%d: %9.1f us  idx %d  %s\n", i, us, hipblaslt_ext::getIndexFromAlgo(heur2[i].algo),
             tile_of(hipblaslt_ext::getSolutionNameFromAlgo(h, heur2[i].algo)).c_str());
    }
    for (size_t i = 0; i < rows.size() && i < 12; ++i) {
      std::vector<int> want{rows[i].idx};
      std::vector<hipblasLtMatmulHeuristicResult_t> one;
      if (hipblaslt_ext::getAlgosFromIndex(h, want, one) != HIPBLAS_STATUS_SUCCESS || one.empty()) continue;
      size_t ws = 0;
      if (hipblaslt_ext::matmulIsAlgoSupported(h, desc, &alpha, la, lb2, &beta, lc2, lc2, one[0].algo, ws) != HIPBLAS_STATUS_SUCCESS || ws != 0) {
        printf("    M's #%zu (idx %d): not supported / workspace %zu at this M\n", i + 1, rows[i].idx, ws);
        continue;
      }
      const double us = time_algo(one[0].algo, lb2, lc2, 1e30, nullptr);
      printf("    M's #%zu (idx %d): %9.1f us  %+6.1f %% vs this M's heuristic #0\n", i + 1, rows[i].idx, us, first2 > 0 ? 100.0 * (us / first2 - 1.0) : 0.0);
    }
  }
  fflush(stdout);
  return 0;
}
