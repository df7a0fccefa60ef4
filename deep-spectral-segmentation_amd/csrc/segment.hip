// segment.hip - the immediate consumers of the eigenvectors ON THE DEVICE, right after the solve:
//   * dss_fiedler_mask     extract/extract.py:383-407  single-region mask  eigenvectors[index] > threshold  (0 / 255)
//   * dss_kmeans_segments  extract/extract.py:283-352  multi-region labels: K-means over the rows of V[first : first + dims]^T
//                          (one point per patch), then the border vote of extract_utils.py:124-135 and the label swap that
//                          makes the segment owning most border patches 0.
// Tiny per-image work (N <= a few thousand points in <= 61 dimensions): one 256-thread workgroup per image, everything in
// LDS, no tuning beyond that.  The reference clusters with sklearn.cluster.KMeans on the host (the CLI commands of
// extract.py keep doing exactly that, bit-identical to the reference for a seeded run); this is the same ALGORITHM -
// Lloyd iterations from given or k-means++ centres, sklearn's two stopping rules, a final assignment - for pipelines that
// want the segmentation without the .pth round trip.  Arithmetic: distances as fp32 sums of squared differences, centroid
// sums in fp64 in a fixed order (deterministic: a (cluster, dimension) pair is owned by one thread that walks the points).
#include "common.h"

namespace dss {

static constexpr int KM_THREADS = 256;
static constexpr int KM_MAX_K = 32;      // clusters
static constexpr int KM_MAX_DIMS = 64;   // coordinates per point (eigenvectors used)
static constexpr int KM_MAX_N = 8192;    // points per image (labels + D^2 live in LDS: 5 bytes per point, < 64 KB in all)

__global__ __launch_bounds__(256) void fiedler_mask_kernel(const float* __restrict__ vec, uint8_t* __restrict__ mask, int K,
                                                           int N, int index, float threshold) {
  const float* v = vec + ((size_t)blockIdx.y * K + index) * N;
  uint8_t* m = mask + (size_t)blockIdx.y * N;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N; e += gridDim.x * blockDim.x) m[e] = v[e] > threshold ? 255 : 0;
}

// counter-based generator for the k-means++ draws: uniform in [0, 1)
__device__ __forceinline__ float km_uniform(unsigned seed, unsigned image, unsigned draw) {
  unsigned h = seed * 0x9E3779B9u + image * 0x85EBCA6Bu + draw * 0xC2B2AE35u + 0x27D4EB2Fu;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float km_block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < KM_THREADS / 64; ++w) t += red[w];
  return t;
}

__global__ __launch_bounds__(KM_THREADS) void kmeans_segments_kernel(
    const float* __restrict__ vec, int K, int N, int first, int dims, int k, const float* __restrict__ init,
    unsigned seed, int max_iter, float tol, int hp, int wp, int infer_bg, uint8_t* __restrict__ labels_out,
    float* __restrict__ inertia_out, int* __restrict__ iters_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* cen = reinterpret_cast<float*>(lds);                       // [k][dims]
  float* cnew = cen + KM_MAX_K * KM_MAX_DIMS;                       // [k][dims]
  float* d2 = cnew + KM_MAX_K * KM_MAX_DIMS;                        // [N]  (k-means++ only; reused for nothing else)
  float* red = d2 + N;                                              // [4]
  int* icnt = reinterpret_cast<int*>(red + 8);                      // [KM_MAX_K] counts / votes
  int* flag = icnt + KM_MAX_K;                                      // [4] scratch scalars
  uint8_t* lab = reinterpret_cast<uint8_t*>(flag + 4);              // [N]
  const int tid = threadIdx.x, img = blockIdx.x;
  const float* X = vec + ((size_t)img * K + first) * N;             // coordinate j of point e: X[j * N + e]

  // ---- tolerance of sklearn's centre-shift rule: tol * mean over the coordinates of their variance ---------------------
  float var_sum = 0.f;
  for (int j = 0; j < dims; ++j) {
    float s = 0.f, s2 = 0.f;
    for (int e = tid; e < N; e += KM_THREADS) { const float x = X[(size_t)j * N + e]; s += x; s2 += x * x; }
    s = km_block_sum(s, red);
    s2 = km_block_sum(s2, red);
    const float mean = s / N;
    var_sum += s2 / N - mean * mean;
  }
  const float tol_abs = tol * var_sum / dims;

  // ---- initial centres: given, or k-means++ (D^2 sampling, one trial per centre) -----------------------------------------
  if (init) {
    for (int i = tid; i < k * dims; i += KM_THREADS) cen[(i / dims) * KM_MAX_DIMS + i % dims] = init[(size_t)img * k * dims + i];
    __syncthreads();
  } else {
    for (int c = 0; c < k; ++c) {
      if (c == 0) {
        if (tid == 0) flag[0] = min(N - 1, (int)(km_uniform(seed, img, 0) * N));
      } else {
        float part = 0.f;
        for (int e = tid; e < N; e += KM_THREADS) part += d2[e];
        const float total = km_block_sum(part, red);
        if (tid == 0) {                                              // serial scan: N <= 8192 steps, a few microseconds
          const float r = km_uniform(seed, img, c) * total;
          float acc = 0.f;
          int pick = N - 1;
          for (int e = 0; e < N; ++e) { acc += d2[e]; if (acc > r) { pick = e; break; } }
          flag[0] = pick;
        }
      }
      __syncthreads();
      const int pick = flag[0];
      for (int j = tid; j < dims; j += KM_THREADS) cen[c * KM_MAX_DIMS + j] = X[(size_t)j * N + pick];
      __syncthreads();
      for (int e = tid; e < N; e += KM_THREADS) {
        float d = 0.f;
        for (int j = 0; j < dims; ++j) { const float t = X[(size_t)j * N + e] - cen[c * KM_MAX_DIMS + j]; d += t * t; }
        d2[e] = c == 0 ? d : fminf(d2[e], d);
      }
      __syncthreads();
    }
  }

  // ---- Lloyd ------------------------------------------------------------------------------------------------------------
  auto assign = [&](bool count_changes) -> int {
    int changed = 0;
    for (int e = tid; e < N; e += KM_THREADS) {
      float best = 3.4e38f;
      int bl = 0;
      for (int c = 0; c < k; ++c) {
        float d = 0.f;
        for (int j = 0; j < dims; ++j) { const float t = X[(size_t)j * N + e] - cen[c * KM_MAX_DIMS + j]; d += t * t; }
        if (d < best) { best = d; bl = c; }                          // ties: the lower label, like argmin
      }
      if (count_changes && lab[e] != (uint8_t)bl) ++changed;
      lab[e] = (uint8_t)bl;
    }
    return changed;
  };
  for (int e = tid; e < N; e += KM_THREADS) lab[e] = 255;
  __syncthreads();
  int it = 0;
  for (; it < max_iter; ++it) {
    const int changed = (int)(km_block_sum((float)assign(true), red) + 0.5f);
    __syncthreads();
    // new centres: thread (c, j) walks all points in order (fp64 sums: deterministic and exact enough to be order-free)
    for (int i = tid; i < k * dims; i += KM_THREADS) {
      const int c = i / dims, j = i % dims;
      double s = 0.0;
      int n = 0;
      for (int e = 0; e < N; ++e)
        if (lab[e] == c) { s += (double)X[(size_t)j * N + e]; ++n; }
      cnew[c * KM_MAX_DIMS + j] = n > 0 ? (float)(s / n) : cen[c * KM_MAX_DIMS + j];   // an empty cluster keeps its centre
    }
    __syncthreads();
    float shift = 0.f;
    for (int i = tid; i < k * dims; i += KM_THREADS) {
      const int c = i / dims, j = i % dims;
      const float t = cnew[c * KM_MAX_DIMS + j] - cen[c * KM_MAX_DIMS + j];
      shift += t * t;
    }
    shift = km_block_sum(shift, red);
    for (int i = tid; i < k * dims; i += KM_THREADS) cen[(i / dims) * KM_MAX_DIMS + i % dims] = cnew[(i / dims) * KM_MAX_DIMS + i % dims];
    __syncthreads();
    if (changed == 0) { ++it; break; }                               // strict convergence: labels are already final
    if (shift <= tol_abs) { ++it; assign(false); __syncthreads(); break; }   // centre shift below tolerance: final E-step
  }

  // ---- inertia -------------------------------------------------------------------------------------------------------------
  float part = 0.f;
  for (int e = tid; e < N; e += KM_THREADS) {
    const int c = lab[e];
    for (int j = 0; j < dims; ++j) { const float t = X[(size_t)j * N + e] - cen[c * KM_MAX_DIMS + j]; part += t * t; }
  }
  const float inertia = km_block_sum(part, red);

  // ---- border vote + label swap (extract_utils.py:124-135, extract.py:343-350): the four border lines are counted
  //      separately (corners twice); the winner - the lowest label among equals, as argmax over ascending labels -
  //      trades places with label 0.
  int bg = 0;
  if (infer_bg && hp * wp == N) {
    for (int c = tid; c < KM_MAX_K; c += KM_THREADS) icnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < 2 * (hp + wp); i += KM_THREADS) {
      int r, cidx;
      if (i < hp) { r = i; cidx = 0; }
      else if (i < 2 * hp) { r = i - hp; cidx = wp - 1; }
      else if (i < 2 * hp + wp) { r = 0; cidx = i - 2 * hp; }
      else { r = hp - 1; cidx = i - 2 * hp - wp; }
      atomicAdd(&icnt[lab[r * wp + cidx]], 1);
    }
    __syncthreads();
    int best = -1;
    for (int c = 0; c < k; ++c)
      if (icnt[c] > best) { best = icnt[c]; bg = c; }
  }
  uint8_t* out = labels_out + (size_t)img * N;
  for (int e = tid; e < N; e += KM_THREADS) {
    const int c = lab[e];
    out[e] = (uint8_t)(c == bg ? 0 : (c == 0 ? bg : c));
  }
  if (tid == 0) {
    inertia_out[img] = inertia;
    iters_out[img] = it;
  }
}

}  // namespace dss

extern "C" int dss_fiedler_mask(const float* eigenvectors, uint8_t* mask, int B, int K, int N, int index, float threshold,
                                void* stream) {
  DSS_REQUIRE(eigenvectors && mask, "dss_fiedler_mask: null pointer");
  DSS_REQUIRE(B > 0 && K > 0 && N > 0 && index >= 0 && index < K, "dss_fiedler_mask: bad shape B=%d K=%d N=%d index=%d", B, K,
              N, index);
  hipLaunchKernelGGL(dss::fiedler_mask_kernel, dim3((unsigned)dss::ceil_div(N, 256), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, eigenvectors, mask, K, N, index, threshold);
  DSS_CHECK_LAUNCH("fiedler_mask");
  return DSS_OK;
}

extern "C" int dss_kmeans_segments(const float* eigenvectors, int B, int K, int N, int first, int dims, int k,
                                   const float* centroids_init, unsigned seed, int max_iter, float tol, int hp, int wp,
                                   int infer_bg, uint8_t* labels, float* inertia, int32_t* iters, void* stream) {
  DSS_REQUIRE(eigenvectors && labels && inertia && iters, "dss_kmeans_segments: null pointer");
  DSS_REQUIRE(B > 0 && N > 0 && N <= dss::KM_MAX_N, "dss_kmeans_segments: need 0 < N <= %d (N=%d)", dss::KM_MAX_N, N);
  DSS_REQUIRE(first >= 0 && dims > 0 && dims <= dss::KM_MAX_DIMS && first + dims <= K,
              "dss_kmeans_segments: eigenvectors [%d, %d) of K=%d (at most %d coordinates)", first, first + dims, K,
              dss::KM_MAX_DIMS);
  DSS_REQUIRE(k > 0 && k <= dss::KM_MAX_K && k <= N, "dss_kmeans_segments: need 0 < k <= min(%d, N) (k=%d)", dss::KM_MAX_K, k);
  DSS_REQUIRE(max_iter > 0 && tol >= 0.f, "dss_kmeans_segments: max_iter=%d tol=%g", max_iter, (double)tol);
  DSS_REQUIRE(!infer_bg || hp * wp == N, "dss_kmeans_segments: border vote needs hp * wp == N (%d x %d vs %d)", hp, wp, N);
  const size_t lds = (size_t)2 * dss::KM_MAX_K * dss::KM_MAX_DIMS * 4 + (size_t)N * 4 + 32 + dss::KM_MAX_K * 4 + 16 + (size_t)N;
  hipLaunchKernelGGL(dss::kmeans_segments_kernel, dim3((unsigned)B), dim3(dss::KM_THREADS), lds, (hipStream_t)stream,
                     eigenvectors, K, N, first, dims, k, centroids_init, seed, max_iter, tol, hp, wp, infer_bg, labels,
                     inertia, iters);
  DSS_CHECK_LAUNCH("kmeans_segments");
  return DSS_OK;
}
