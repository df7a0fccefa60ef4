// preprocess.hip - image transform kernels.
//
// Replaces (reference, CPU dataloader workers):
//   extract/extract_utils.py:55-56  transforms.ToTensor() + Normalize(ImageNet mean/std)
//   extract/extract.py:82-88        crop to a multiple of the patch size (top-left)
// plus the im2col view of DINO's PatchEmbed Conv2d(3, D, P, P) (kernel == stride, so im2col is a
// pure re-indexing) so the patch embedding becomes one dense GEMM.
//
// HBM-bound byte shuffling: 1 B read, 2-4 B written per element.  Writes are the wide side, so
// threads are mapped to consecutive OUTPUT elements (coalesced stores); reads are strided by 3 B
// within a row of pixels and come out of L1/L2.
#include "common.h"

namespace dss {

__device__ __forceinline__ float transform_px(uint8_t v, int c) {
  // ToTensor: float(v) / 255 ; Normalize: (x - mean[c]) / std[c]   (IEEE fp32 division, no fast-math)
  const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
  const float stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
  float x = (float)v / 255.0f;
  return (x - mean) / stdv;
}

__global__ void preprocess_chw_kernel(const uint8_t* __restrict__ img, float* __restrict__ out,
                                      int H, int W, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const long hw = (long)H * W;
  for (; i < total; i += stride) {
    long b = i / (3 * hw);
    long r = i - b * 3 * hw;
    int c = (int)(r / hw);
    long p = r - (long)c * hw;  // y*W + x
    out[i] = transform_px(img[(b * hw + p) * 3 + c], c);
  }
}

template <class T>
__global__ void preprocess_patchify_kernel(const uint8_t* __restrict__ img, T* __restrict__ out,
                                           int H, int W, int P, int Hp, int Wp, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const int pp = P * P;
  const int inner = 3 * pp;
  const long per_img = (long)Hp * Wp * inner;
  for (; i < total; i += stride) {
    long b = i / per_img;
    long r = i - b * per_img;
    int n = (int)(r / inner);
    int q = (int)(r - (long)n * inner);
    int c = q / pp;
    int s = q - c * pp;
    int py = s / P, px = s - py * P;
    int y = (n / Wp) * P + py;
    int x = (n % Wp) * P + px;
    out[i] = from_f32<T>(transform_px(img[((b * H + y) * (long)W + x) * 3 + c], c));
  }
}

// Fast path for P % 8 == 0 (every DINO ViT): one thread = 8 consecutive pixels of one patch row, all 3 channels:
// 24 contiguous input bytes -> three 16-byte stores (8 halves / floats of one (c, py) segment).  The index
// arithmetic (the expensive part of the generic kernel) is paid once per 24 elements.
// The transform has 3 x 256 possible results: every workgroup tabulates them once (the two IEEE divisions of
// transform_px are ~20 instructions per element, 480 of the kernel's 960 per chunk - it ran at 3 TB/s, issue-bound) and
// an element becomes one LDS read of the SAME value, already rounded to T.
template <class T>
__global__ __launch_bounds__(256) void preprocess_patchify8_kernel(const uint8_t* __restrict__ img, T* __restrict__ out, int H, int W,
                                                                   int P, int Hp, int Wp, long total_chunks) {
  __shared__ T lut[3][256];
  for (int e = threadIdx.x; e < 768; e += blockDim.x) lut[e >> 8][e & 255] = from_f32<T>(transform_px((uint8_t)(e & 255), e >> 8));
  __syncthreads();
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const int cpr = P >> 3;                 // chunks per patch row
  const int cpp = P * cpr;                // chunks per patch
  const long per_img = (long)Hp * Wp * cpp;
  for (; i < total_chunks; i += stride) {
    const long b = i / per_img;
    const int r = (int)(i - b * per_img);
    const int n = r / cpp;
    const int q = r - n * cpp;
    const int py = q / cpr, xc = q - py * cpr;
    const int y = (n / Wp) * P + py;
    const int x = (n % Wp) * P + 8 * xc;
    const uint8_t* src = img + ((b * H + y) * (long)W + x) * 3;
    uint8_t px[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) px[k] = src[k];
    T* dst = out + ((b * Hp * Wp + n) * 3L) * P * P + py * P + 8 * xc;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      T v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = lut[c][px[3 * k + c]];
      T* d = dst + (long)c * P * P;
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = v[k];
    }
  }
}

template <class T>
static void launch_patchify(const uint8_t* img, void* out, int B, int H, int W, int P, int Hp, int Wp, hipStream_t s) {
  const int threads = 256;
  if ((P & 7) == 0) {
    const long chunks = (long)B * Hp * Wp * P * (P >> 3);
    long blocks = (chunks + threads - 1) / threads;
    if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(preprocess_patchify8_kernel<T>, dim3((unsigned)blocks), dim3(threads), 0, s, img, (T*)out, H,
                       W, P, Hp, Wp, chunks);
  } else {
    const long total = (long)B * Hp * Wp * 3 * P * P;
    long blocks = (total + threads - 1) / threads;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(preprocess_patchify_kernel<T>, dim3((unsigned)blocks), dim3(threads), 0, s, img, (T*)out, H, W,
                       P, Hp, Wp, total);
  }
}

}  // namespace dss

extern "C" int dss_preprocess_chw(const uint8_t* img_u8, float* out_chw, int B, int H, int W,
                                  void* stream) {
  DSS_REQUIRE(img_u8 && out_chw, "dss_preprocess_chw: null pointer");
  DSS_REQUIRE(B > 0 && H > 0 && W > 0, "dss_preprocess_chw: bad shape B=%d H=%d W=%d", B, H, W);
  const long total = (long)B * 3 * H * W;
  const int threads = 256;
  long blocks = (total + threads - 1) / threads;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dss::preprocess_chw_kernel, dim3((unsigned)blocks), dim3(threads), 0,
                     (hipStream_t)stream, img_u8, out_chw, H, W, total);
  DSS_CHECK_LAUNCH("preprocess_chw");
  return DSS_OK;
}

extern "C" int dss_preprocess_patchify(const uint8_t* img_u8, void* out, int B, int H, int W, int P,
                                       int out_dtype, void* stream) {
  DSS_REQUIRE(img_u8 && out, "dss_preprocess_patchify: null pointer");
  DSS_REQUIRE(B > 0 && H > 0 && W > 0 && P > 0, "dss_preprocess_patchify: bad shape");
  const int Hp = H / P, Wp = W / P;
  DSS_REQUIRE(Hp > 0 && Wp > 0, "dss_preprocess_patchify: image %dx%d smaller than patch %d", H, W, P);
  hipStream_t s = (hipStream_t)stream;
  switch (out_dtype) {
    case DSS_F32: dss::launch_patchify<float>(img_u8, out, B, H, W, P, Hp, Wp, s); break;
    case DSS_F16: dss::launch_patchify<dss::f16>(img_u8, out, B, H, W, P, Hp, Wp, s); break;
    case DSS_BF16: dss::launch_patchify<dss::bf16>(img_u8, out, B, H, W, P, Hp, Wp, s); break;
    default:
      return dss::fail(DSS_ERR_BAD_ARG, "dss_preprocess_patchify: unsupported out_dtype %d", out_dtype);
  }
  DSS_CHECK_LAUNCH("preprocess_patchify");
  return DSS_OK;
}
