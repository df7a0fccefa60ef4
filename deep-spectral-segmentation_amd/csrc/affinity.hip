// affinity.hip - row L2 normalisation and the patch-feature affinity W = relu(F F^T), exact fp32 on MFMA.
//
// Replaces (reference, torch ops on the GPU followed by an N^2 device->host copy):
//   extract/extract.py:148      feats = F.normalize(feats, p=2, dim=-1)
//   extract/extract.py:191-193  W_feat = feats @ feats.T ; W_feat = W_feat * (W_feat > 0)
//   extract/extract.py:194      W_feat / W_feat.max()   -> dropped: (D-W)v = lambda D v is scale invariant
//   extract/extract.py:195      .cpu().numpy()          -> eliminated: W never leaves HBM
//
// Gram kernel: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bitwise an fmaf chain, no TF32-style
// truncation - gfx950 has none).  Roofline: 2*N^2*D flop against 157.3 TF/s fp32-MFMA; arithmetic
// intensity D/2 flop/B >> the fp32 ridge (~20), so this kernel is MFMA-bound, not HBM-bound.
// Tiling: 256 threads = 4 waves (2x2), block tile 128x128, each wave 64x64 = 2x2 MFMA tiles (64
// accumulator registers); F panels [128 rows x 32 floats] staged through LDS with a 4-float row skew
// so that ds_read_b128 fragment reads are bank-conflict free (row stride 36 words: 16 rows cover the 16
// four-bank slots exactly once).
//
// MFMA operand maps (cdna_hip_programming.md §3): A: lane l holds A[i=l&31][k=l>>5]; B: B[k=l>>5][j=l&31];
// D: register r of lane l is D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31].  The k index is a dummy: lane half hh
// feeds feature columns d0+4*hh+s to MFMA number s for both operands.
#include "common.h"
#include "eigs_core.h"

namespace dss {

// ------------------------------------------------------------------------------------------------
// y = x / max(||x||, eps) per row; one wave per row.
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* xr = x + row * D;
    float* yr = y + row * D;
    float ss = 0.f;
    if ((D & 3) == 0) {
      for (int c = lane; c < (D >> 2); c += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    } else {
      for (int c = lane; c < D; c += 64) ss += xr[c] * xr[c];
    }
    const float denom = fmaxf(sqrtf(wave_sum(ss)), eps);
    if ((D & 3) == 0) {
      for (int c = lane; c < (D >> 2); c += 64) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        v[0] /= denom; v[1] /= denom; v[2] /= denom; v[3] /= denom;
        *reinterpret_cast<f32x4*>(yr + 4 * c) = v;
      }
    } else {
      for (int c = lane; c < D; c += 64) yr[c] = xr[c] / denom;
    }
  }
}

// ------------------------------------------------------------------------------------------------
static constexpr int GB = 128;  // block tile (rows and cols)

// Index `rem` of a block inside one image's upper block triangle (nbk x nbk blocks of GB rows, bj >= bi) -> (bi, bj),
// enumerated in SUPER-TILES of GST x GST blocks: the workgroups of an image that run at the same time (consecutive
// indices on one XCD) then touch ~2 GST panels per GST^2 blocks instead of a whole block row's worth.  At N = 900 an
// image's features fit an XCD's L2 either way; at N = 3600, D = 768 (11 MB per image) the plain row-major order re-read
// its panels 5.8x through the fabric (profiles/r02_c3_pmc_traffic.json).  All-scalar arithmetic, <= ~40 iterations.
static constexpr int GST = 4;
__device__ __forceinline__ void gram_block_of(int rem, int nbk, int& bi, int& bj) {
  const int ns = (nbk + GST - 1) / GST;
  for (int si = 0; si < ns; ++si) {
    const int r0 = si * GST, r1 = min(r0 + GST, nbk);
    for (int sj = si; sj < ns; ++sj) {
      const int c0 = sj * GST, c1 = min(c0 + GST, nbk);
      const int h = r1 - r0, w = c1 - c0;
      const int cnt = sj == si ? h * (h + 1) / 2 : h * w;
      if (rem < cnt) {
        if (sj == si) {                       // diagonal super-tile: its own little upper triangle
          int i = 0;
          while (rem >= h - i) { rem -= h - i; ++i; }
          bi = r0 + i;
          bj = r0 + i + rem;
        } else {
          bi = r0 + rem / w;
          bj = c0 + rem % w;
        }
        return;
      }
      rem -= cnt;
    }
  }
  bi = bj = 0;                                // not reached for rem < nbk (nbk + 1) / 2
}
static constexpr int GK = 32;   // feature columns per LDS stage
static constexpr int GLD = GK + 4;

// Symmetric output: only block tiles with J0 >= I0 are computed (1-D grid over the upper block triangle) and each
// wave's 64x64 quadrant is written as ONE packed storage tile (eigs_core.h: wsym_*), so the Gram FLOPs and the
// bytes the eigensolver later streams are both halved.  Quadrants below the diagonal are skipped.
__global__ __launch_bounds__(256) void gram_relu_kernel(const float* __restrict__ feats, float* __restrict__ W,
                                                        int N, int D, int ldw, int relu, size_t w_stride,
                                                        int nimg) {
  typedef float WE;   // storage type of W (the split-f16 kernel below also writes the u16 form)
  __shared__ __attribute__((aligned(16))) float As[GB][GLD];
  __shared__ __attribute__((aligned(16))) float Bs[GB][GLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int nt = ldw / 64, nbk = (nt + 1) / 2, nblk = nbk * (nbk + 1) / 2;
  // XCD-aware 1-D block order: workgroup id % 8 selects the XCD (observed), so all blocks of one image get ids
  // congruent mod 8 and its feature panels (N*D*4 B, L2-sized) are fetched from HBM once instead of once per XCD
  // (PMC before: 10.2 GB fetched for 1.4 GB of features).  Speed heuristic only.
  int img, rem;
  {
    const int id = blockIdx.x, g8 = nimg & ~7;
    if (id < nblk * g8) {
      const int xcd = id & 7, slot = id >> 3;
      img = (slot / nblk) * 8 + xcd;
      rem = slot % nblk;
    } else {
      const int r = id - nblk * g8;
      img = g8 + r / nblk;
      rem = r % nblk;
    }
  }
  int bi, bj;                                        // upper-triangular block index -> (bi, bj), bj >= bi
  gram_block_of(rem, nbk, bi, bj);
  const int I0 = bi * GB, J0 = bj * GB;
  const float* F = feats + (long)img * N * D;
  float* Wb = W + img * w_stride;

  // this wave's 64x64 quadrant = storage tile (ti, tj); active only if inside the matrix and on/above the diagonal
  const int ti = 2 * bi + wr, tj = 2 * bj + wc;
  const bool quad = ti < nt && tj < nt && tj >= ti;
  const int ri0 = I0 + wr * 64, cj0 = J0 + wc * 64;
  const bool act_r0 = quad && ri0 < N, act_r1 = quad && ri0 + 32 < N;
  const bool act_c0 = quad && cj0 < N, act_c1 = quad && cj0 + 32 < N;

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  // staging map: 8 threads cover one 128-byte row segment; 32 rows per pass, 4 passes per panel
  const int srow = tid >> 3, scol = (tid & 7) * 4;
  for (int d0 = 0; d0 < D; d0 += GK) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = srow + 32 * p;
      int ga = I0 + r; ga = ga < N ? ga : N - 1;  // clamp: rows past N are computed and discarded
      int gb = J0 + r; gb = gb < N ? gb : N - 1;
      *reinterpret_cast<f32x4*>(&As[r][scol]) = *reinterpret_cast<const f32x4*>(F + (long)ga * D + d0 + scol);
      *reinterpret_cast<f32x4*>(&Bs[r][scol]) = *reinterpret_cast<const f32x4*>(F + (long)gb * D + d0 + scol);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK / 8; ++kk) {
      const int c = 8 * kk + 4 * hh;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&As[wr * 64 + li][c]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&As[wr * 64 + 32 + li][c]);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bs[wc * 64 + li][c]);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Bs[wc * 64 + 32 + li][c]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (act_r0 && act_c0) acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc00, 0, 0, 0);
        if (act_r0 && act_c1) acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc01, 0, 0, 0);
        if (act_r1 && act_c0) acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc10, 0, 0, 0);
        if (act_r1 && act_c1) acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc11, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue: relu; rows/columns >= N are written as 0 (the matvec relies on it); each half-wave stores 32
  // consecutive floats (128 B) of a tile row.  The whole 64x64 storage tile is always written.
  if (!quad) return;
  WE* tile = Wb + (size_t)(wsym_row_start(ti, nt) + (tj - ti)) * 64 * 64;
  auto store_tile = [&](const f32x16& acc, int rsub, int csub, bool computed) {
    const int col = cj0 + csub + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = rsub + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const int row = ri0 + lr;
      float v = computed ? acc[r] : 0.f;
      if (relu) v = fmaxf(v, 0.f);
      if (row >= N || col >= N) v = 0.f;
      if constexpr (sizeof(WE) == 4) {
        tile[lr * 64 + csub + li] = v;
      } else {
        const float q = __builtin_rintf(fminf(fmaxf(v, 0.f), 1.0f) * 65535.0f);
        tile[lr * 64 + csub + li] = (WE)(unsigned)q;
      }
    }
  };
  store_tile(acc00, 0, 0, act_r0 && act_c0);
  store_tile(acc01, 0, 32, act_r0 && act_c1);
  store_tile(acc10, 32, 0, act_r1 && act_c0);
  store_tile(acc11, 32, 32, act_r1 && act_c1);
}

// ================================================================================================
// Split-f16 affinity: fp32-class accuracy at the f16 MFMA rate.
//   x = hi + lo,  hi = f16(x),  lo = f16((x - hi) * 2^6) / 2^6          (|x| <= 1 after normalisation)
//   <x_i, x_j> = hi_i.hi_j + (hi_i 2^-6).(lo_j 2^6) + (lo_i 2^6).(hi_j 2^-6) + O(2^-22)
// Every f16 x f16 product is exact in the fp32 accumulator, the 2^+-6 scalings keep hi*2^-6 and lo*2^6 in the
// normal f16 range (no subnormals), and all three products accumulate into ONE v_mfma_f32_32x32x16_f16
// accumulator.  3 MFMAs of 32 cycles per 16 feature columns instead of 8 fp32 MFMAs of 64 cycles: the Gram
// stops being MFMA-bound and sits on the HBM roofline (bytes: 4 B/elem of features in, 4 B/elem of packed W out).
__global__ __launch_bounds__(256) void normalize_rows_split_kernel(const float* __restrict__ x, f16* __restrict__ hi,
                                                                   f16* __restrict__ lo, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* xr = x + row * D;
    float ss = 0.f;
    for (int c = lane; c < (D >> 2); c += 64) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c);
      ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float denom = eps < 0.f ? 1.0f : fmaxf(sqrtf(wave_sum(ss)), eps);
    for (int c = lane; c < (D >> 2); c += 64) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 4 * c);
      f16x4 h, l;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float y = v[i] / denom;
        h[i] = (f16)y;
        l[i] = (f16)((y - (float)h[i]) * 64.0f);
      }
      *reinterpret_cast<f16x4*>(hi + row * D + 4 * c) = h;
      *reinterpret_cast<f16x4*>(lo + row * D + 4 * c) = l;
    }
  }
}

static constexpr int SK = 32;        // feature columns per LDS stage (two k16 MFMA steps)
static constexpr int SLD = SK + 8;   // 80-byte rows: 16 rows land on 16 distinct 16-byte bank slots

// WE = float: W as is.  WE = uint16_t: round(65535 w) clamped to [0, 65535] (relu implied; see eigs_core.h WElem).
template <class WE>
__global__ __launch_bounds__(256) void gram_split_kernel(const f16* __restrict__ Hi, const f16* __restrict__ Lo,
                                                         WE* __restrict__ W, int N, int D, int ldw, int relu,
                                                         size_t w_stride, int nimg) {
  __shared__ __attribute__((aligned(16))) f16 Ah[GB][SLD];
  __shared__ __attribute__((aligned(16))) f16 Al[GB][SLD];
  __shared__ __attribute__((aligned(16))) f16 Bh[GB][SLD];
  __shared__ __attribute__((aligned(16))) f16 Bl[GB][SLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int nt = ldw / 64, nbk = (nt + 1) / 2, nblk = nbk * (nbk + 1) / 2;
  int img, rem;
  {  // XCD-aware order: all blocks of one image share an XCD (see gram_relu_kernel)
    const int id = blockIdx.x, g8 = nimg & ~7;
    if (id < nblk * g8) {
      const int xcd = id & 7, slot = id >> 3;
      img = (slot / nblk) * 8 + xcd;
      rem = slot % nblk;
    } else {
      const int r = id - nblk * g8;
      img = g8 + r / nblk;
      rem = r % nblk;
    }
  }
  int bi, bj;
  gram_block_of(rem, nbk, bi, bj);
  const int I0 = bi * GB, J0 = bj * GB;
  const f16* Hb = Hi + (long)img * N * D;
  const f16* Lb = Lo + (long)img * N * D;
  WE* Wb = W + img * w_stride;

  const int ti = 2 * bi + wr, tj = 2 * bj + wc;
  const bool quad = ti < nt && tj < nt && tj >= ti;
  const int ri0 = I0 + wr * 64, cj0 = J0 + wc * 64;
  const bool act_r0 = quad && ri0 < N, act_r1 = quad && ri0 + 32 < N;
  const bool act_c0 = quad && cj0 < N, act_c1 = quad && cj0 + 32 < N;

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  // staging: a [128 x 32] f16 panel is 512 16-byte chunks; thread -> chunks tid and tid + 256
  const int srow = tid >> 2, scol = (tid & 3) * 8;
  const f16 s_dn = (f16)0.015625f;  // 2^-6
  for (int d0 = 0; d0 < D; d0 += SK) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = srow + 64 * p;
      int ga = I0 + r; ga = ga < N ? ga : N - 1;
      int gb = J0 + r; gb = gb < N ? gb : N - 1;
      *reinterpret_cast<f16x8*>(&Ah[r][scol]) = *reinterpret_cast<const f16x8*>(Hb + (long)ga * D + d0 + scol);
      *reinterpret_cast<f16x8*>(&Al[r][scol]) = *reinterpret_cast<const f16x8*>(Lb + (long)ga * D + d0 + scol);
      *reinterpret_cast<f16x8*>(&Bh[r][scol]) = *reinterpret_cast<const f16x8*>(Hb + (long)gb * D + d0 + scol);
      *reinterpret_cast<f16x8*>(&Bl[r][scol]) = *reinterpret_cast<const f16x8*>(Lb + (long)gb * D + d0 + scol);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < SK / 16; ++kk) {
      const int c = 16 * kk + 8 * hh;
      const f16x8 ah0 = *reinterpret_cast<const f16x8*>(&Ah[wr * 64 + li][c]);
      const f16x8 ah1 = *reinterpret_cast<const f16x8*>(&Ah[wr * 64 + 32 + li][c]);
      const f16x8 al0 = *reinterpret_cast<const f16x8*>(&Al[wr * 64 + li][c]);
      const f16x8 al1 = *reinterpret_cast<const f16x8*>(&Al[wr * 64 + 32 + li][c]);
      const f16x8 bh0 = *reinterpret_cast<const f16x8*>(&Bh[wc * 64 + li][c]);
      const f16x8 bh1 = *reinterpret_cast<const f16x8*>(&Bh[wc * 64 + 32 + li][c]);
      const f16x8 bl0 = *reinterpret_cast<const f16x8*>(&Bl[wc * 64 + li][c]);
      const f16x8 bl1 = *reinterpret_cast<const f16x8*>(&Bl[wc * 64 + 32 + li][c]);
      const f16x8 as0 = ah0 * s_dn, as1 = ah1 * s_dn, bs0 = bh0 * s_dn, bs1 = bh1 * s_dn;  // exact exponent shifts
      if (act_r0 && act_c0) {
        acc00 = mfma32x32x16(ah0, bh0, acc00); acc00 = mfma32x32x16(as0, bl0, acc00); acc00 = mfma32x32x16(al0, bs0, acc00);
      }
      if (act_r0 && act_c1) {
        acc01 = mfma32x32x16(ah0, bh1, acc01); acc01 = mfma32x32x16(as0, bl1, acc01); acc01 = mfma32x32x16(al0, bs1, acc01);
      }
      if (act_r1 && act_c0) {
        acc10 = mfma32x32x16(ah1, bh0, acc10); acc10 = mfma32x32x16(as1, bl0, acc10); acc10 = mfma32x32x16(al1, bs0, acc10);
      }
      if (act_r1 && act_c1) {
        acc11 = mfma32x32x16(ah1, bh1, acc11); acc11 = mfma32x32x16(as1, bl1, acc11); acc11 = mfma32x32x16(al1, bs1, acc11);
      }
    }
    __syncthreads();
  }

  if (!quad) return;
  WE* tile = Wb + (size_t)(wsym_row_start(ti, nt) + (tj - ti)) * 64 * 64;
  auto store_tile = [&](const f32x16& acc, int rsub, int csub, bool computed) {
    const int col = cj0 + csub + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = rsub + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const int row = ri0 + lr;
      float v = computed ? acc[r] : 0.f;
      if (relu) v = fmaxf(v, 0.f);
      if (row >= N || col >= N) v = 0.f;
      if constexpr (sizeof(WE) == 4) {
        tile[lr * 64 + csub + li] = v;
      } else {
        const float q = __builtin_rintf(fminf(fmaxf(v, 0.f), 1.0f) * 65535.0f);
        tile[lr * 64 + csub + li] = (WE)(unsigned)q;
      }
    }
  };
  store_tile(acc00, 0, 0, act_r0 && act_c0);
  store_tile(acc01, 0, 32, act_r0 && act_c1);
  store_tile(acc10, 32, 0, act_r1 && act_c0);
  store_tile(acc11, 32, 32, act_r1 && act_c1);
}


// ================================================================================================
// Fused affinity build of the DEFAULT recipe (normalize=True, threshold_at_zero=True, W as 16-bit fixed point):
//   extract/extract.py:148      feats = F.normalize(feats, p=2, dim=-1)
//   extract/extract.py:191-193  W = feats @ feats.T ; W = W * (W > 0)
// in ONE kernel that reads the RAW fp32 features and writes the packed u16 tiles - HBM traffic == the algorithmic bytes
// (4 N D in, N (N+1) out per image), where the split-f16 pair above moves 4 N D three times (normalise + split pass out
// and back in).  Three things make that possible:
//   * normalisation AFTER the product: W_ij = <x_i, x_j> / (|x_i| |x_j|), so a block streams raw panels and collects the
//     squared norms of its own 128 + 128 rows on the way (v_dot2 on the values the MFMAs see, so W_ii = 1 exactly);
//   * f16 operands, ONE v_mfma_f32_32x32x16_f16 per product (fp32 accumulate) instead of the three of the hi/lo split:
//     BASELINE.json config 5 is "fp16 features + fp32 Laplacian eigensolve", the 16-bit W behind it is quantised to 7.6e-6
//     anyway, and the rounding (2^-11 relative per feature: ~3e-5 absolute on w) moves the eigenvectors of the reference
//     goldens by <= 3e-6 in cosine (tests/test_gpu_kernels.py) - the 1e-4 budget is untouched;
//   * same tiling / packed symmetric output / XCD-aware block order as gram_split_kernel.
// LDS: f16 panels [128][32 + 8] (80-byte rows: ds_read_b128 conflict-free) for A and B, + the 256 inverse norms.
// Measured and rejected: 64-column stages in a DOUBLE-buffered LDS image with one barrier per stage (16 MFMAs per wave
// between barriers instead of 8 between two; 220 VGPRs, two workgroups per CU): 2.40 ms vs 2.30 ms per 2030 images - the
// third resident workgroup hides more latency than the second barrier costs.
static constexpr int FK = 32;        // feature columns per stage
static constexpr int FLD = FK + 8;   // halves per LDS row (80 B)

__global__ __launch_bounds__(256, 3) void gram_f16_fused_kernel(const float* __restrict__ feats, uint16_t* __restrict__ W,
                                                                int N, int D, int ldw, float eps, size_t w_stride,
                                                                int nimg) {
  __shared__ __attribute__((aligned(16))) f16 Ah[GB][FLD];
  __shared__ __attribute__((aligned(16))) f16 Bh[GB][FLD];
  __shared__ float rinv[2][GB];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int nt = ldw / 64, nbk = (nt + 1) / 2, nblk = nbk * (nbk + 1) / 2;
  int img, rem;
  {  // XCD-aware order: all blocks of one image share an XCD (see gram_relu_kernel): its panels are L2 hits after one read
    const int id = blockIdx.x, g8 = nimg & ~7;
    if (id < nblk * g8) {
      const int xcd = id & 7, slot = id >> 3;
      img = (slot / nblk) * 8 + xcd;
      rem = slot % nblk;
    } else {
      const int r = id - nblk * g8;
      img = g8 + r / nblk;
      rem = r % nblk;
    }
  }
  int bi, bj;
  gram_block_of(rem, nbk, bi, bj);
  const int I0 = bi * GB, J0 = bj * GB;
  const float* F = feats + (long)img * N * D;
  uint16_t* Wb = W + img * w_stride;

  const int ti = 2 * bi + wr, tj = 2 * bj + wc;
  const bool quad = ti < nt && tj < nt && tj >= ti;
  const int ri0 = I0 + wr * 64, cj0 = J0 + wc * 64;

  // accXY: rows block X (0/1) x columns block Y (0/1) of the wave's 64 x 64 quadrant, computed TRANSPOSED: the MFMA's
  // "A" operand carries the COLUMN panel and its "B" operand the ROW panel, so a lane owns one row of W and its registers
  // run along the columns - 4 consecutive columns per register group = one 8-byte store of four u16 (the untransposed
  // product leaves a lane with one column and 2-byte stores: 64 store instructions per lane instead of 16).
  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  // staging map: 8 threads cover one 128-byte row segment (32 floats); 32 rows per pass, 4 passes per panel.
  // A diagonal block (bi == bj) has ONE panel: its B loads, conversions and LDS image are skipped.
  const bool diag = bi == bj;
  const int srow = tid >> 3, scol = (tid & 7) * 4;
  float ssa[4] = {0.f, 0.f, 0.f, 0.f}, ssb[4] = {0.f, 0.f, 0.f, 0.f};
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  f32x4 va[4], vb[4];
  auto load_chunk = [&](int d0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = srow + 32 * p;
      int ga = I0 + r; ga = ga < N ? ga : N - 1;  // clamp: rows past N are computed and discarded
      va[p] = *reinterpret_cast<const f32x4*>(F + (long)ga * D + d0 + scol);
      if (!diag) {
        int gb = J0 + r; gb = gb < N ? gb : N - 1;
        vb[p] = *reinterpret_cast<const f32x4*>(F + (long)gb * D + d0 + scol);
      }
    }
  };
  load_chunk(0);
  const f16 (*Bp)[FLD] = diag ? Ah : Bh;
  for (int d0 = 0; d0 < D; d0 += FK) {
    // registers -> f16 -> LDS, squared norms of the ROUNDED values on the way
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = srow + 32 * p;
      const h2 a0 = {(f16)va[p][0], (f16)va[p][1]}, a1 = {(f16)va[p][2], (f16)va[p][3]};
      ssa[p] = __builtin_amdgcn_fdot2(a0, a0, ssa[p], false);
      ssa[p] = __builtin_amdgcn_fdot2(a1, a1, ssa[p], false);
      const f16x4 pa = {a0[0], a0[1], a1[0], a1[1]};
      *reinterpret_cast<f16x4*>(&Ah[r][scol]) = pa;
      if (!diag) {
        const h2 b0 = {(f16)vb[p][0], (f16)vb[p][1]}, b1 = {(f16)vb[p][2], (f16)vb[p][3]};
        ssb[p] = __builtin_amdgcn_fdot2(b0, b0, ssb[p], false);
        ssb[p] = __builtin_amdgcn_fdot2(b1, b1, ssb[p], false);
        const f16x4 pb = {b0[0], b0[1], b1[0], b1[1]};
        *reinterpret_cast<f16x4*>(&Bh[r][scol]) = pb;
      }
    }
    __syncthreads();
    if (d0 + FK < D) load_chunk(d0 + FK);      // next chunk's global loads fly under this chunk's MFMAs
    if (quad) {
#pragma unroll
      for (int kk = 0; kk < FK / 16; ++kk) {
        const int c = 16 * kk + 8 * hh;
        const f16x8 r0 = *reinterpret_cast<const f16x8*>(&Ah[wr * 64 + li][c]);        // row panel -> "B" operand
        const f16x8 r1 = *reinterpret_cast<const f16x8*>(&Ah[wr * 64 + 32 + li][c]);
        const f16x8 c0 = *reinterpret_cast<const f16x8*>(&Bp[wc * 64 + li][c]);        // column panel -> "A" operand
        const f16x8 c1 = *reinterpret_cast<const f16x8*>(&Bp[wc * 64 + 32 + li][c]);
        acc00 = mfma32x32x16(c0, r0, acc00);
        acc01 = mfma32x32x16(c1, r0, acc01);
        acc10 = mfma32x32x16(c0, r1, acc10);
        acc11 = mfma32x32x16(c1, r1, acc11);
      }
    }
    __syncthreads();
  }
  // inverse norms of the block's rows: the 8 threads of a row segment hold its partial sums
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float a = ssa[p], b = ssb[p];
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((tid & 7) == 0) {
      const float ra = 1.0f / fmaxf(sqrtf(a), eps);
      rinv[0][srow + 32 * p] = ra;
      rinv[1][srow + 32 * p] = diag ? ra : 1.0f / fmaxf(sqrtf(b), eps);
    }
  }
  __syncthreads();
  if (!quad) return;
  // epilogue: w = <x_i, x_j> / (|x_i| |x_j|); round(65535 clamp(w, 0, 1)) is ONE v_cvt_pknorm_u16_f32 per pair (the clamp
  // is the relu); rows / columns >= N are written as 0 (the matvec relies on it).  The whole storage tile is written.
  uint16_t* tile = Wb + (size_t)(wsym_row_start(ti, nt) + (tj - ti)) * 64 * 64;
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto store_tile = [&](const f32x16& acc, int rsub, int csub) {
    const int lrow = rsub + li;                                             // this lane's row inside the quadrant
    const float rr = ri0 + lrow < N ? rinv[0][wr * 64 + lrow] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int lc = csub + 8 * g + 4 * hh;                                 // 4 consecutive columns lc .. lc + 3
      const f32x4 rc = *reinterpret_cast<const f32x4*>(&rinv[1][wc * 64 + lc]);
      float w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = cj0 + lc + i < N ? acc[4 * g + i] * rr * rc[i] : 0.f;
      const u16x2 q0 = __builtin_amdgcn_cvt_pknorm_u16(w[0], w[1]), q1 = __builtin_amdgcn_cvt_pknorm_u16(w[2], w[3]);
      const u32x2 out = {__builtin_bit_cast(unsigned, q0), __builtin_bit_cast(unsigned, q1)};
      *reinterpret_cast<u32x2*>(tile + lrow * 64 + lc) = out;
    }
  };
  store_tile(acc00, 0, 0);
  store_tile(acc01, 0, 32);
  store_tile(acc10, 32, 0);
  store_tile(acc11, 32, 32);
}

}  // namespace dss

extern "C" int dss_normalize_rows(const float* x, float* y, int rows, int D, float eps, void* stream) {
  DSS_REQUIRE(x && y, "dss_normalize_rows: null pointer");
  DSS_REQUIRE(rows > 0 && D > 0, "dss_normalize_rows: bad shape rows=%d D=%d", rows, D);
  int blocks = dss::ceil_div(rows, 4);
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(dss::normalize_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, rows,
                     D, eps);
  DSS_CHECK_LAUNCH("normalize_rows");
  return DSS_OK;
}

extern "C" int dss_affinity_ld(int N) { return N > 0 ? dss::round_up(N, 64) : 0; }
extern "C" size_t dss_affinity_elems(int N) { return N > 0 ? dss::wsym_floats(dss_affinity_ld(N)) : 0; }

extern "C" int dss_affinity(const float* feats, float* W, int B, int N, int D, int threshold_at_zero,
                            void* stream) {
  DSS_REQUIRE(feats && W, "dss_affinity: null pointer");
  DSS_REQUIRE(B > 0 && N > 0 && D > 0, "dss_affinity: bad shape B=%d N=%d D=%d", B, N, D);
  DSS_REQUIRE(D % dss::GK == 0, "dss_affinity: feature dim must be a multiple of %d (got %d)", dss::GK, D);
  const int ldw = dss_affinity_ld(N);
  const int nbk = (ldw / 64 + 1) / 2;
  const long nblocks = (long)(nbk * (nbk + 1) / 2) * B;
  DSS_REQUIRE(nblocks < 2147483647L, "dss_affinity: too many blocks (%ld)", nblocks);
  hipLaunchKernelGGL(dss::gram_relu_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, feats, W, N,
                     D, ldw, threshold_at_zero ? 1 : 0, dss_affinity_elems(N), B);
  DSS_CHECK_LAUNCH("gram_relu");
  return DSS_OK;
}

namespace dss {
static int affinity_split(const float* feats, void* W, int w_u16, int B, int N, int D, int normalize, float eps,
                          int threshold_at_zero, void* workspace, size_t workspace_bytes, void* stream) {
  DSS_REQUIRE(feats && W && workspace, "dss_affinity_split: null pointer");
  DSS_REQUIRE(B > 0 && N > 0 && D > 0, "dss_affinity_split: bad shape B=%d N=%d D=%d", B, N, D);
  DSS_REQUIRE(D % dss::SK == 0, "dss_affinity_split: feature dim must be a multiple of %d (got %d)", dss::SK, D);
  const size_t need = dss_affinity_split_workspace_bytes(B, N, D);
  if (workspace_bytes < need)
    return dss::fail(DSS_ERR_WORKSPACE, "dss_affinity_split: workspace %zu < %zu bytes", workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  dss::f16* hi = (dss::f16*)workspace;
  dss::f16* lo = hi + (size_t)B * N * D;
  const long rows = (long)B * N;
  int blocks = (int)((rows + 3) / 4 > 65536 ? 65536 : (rows + 3) / 4);
  // normalize == 0: eps is ignored and rows are split as they are (|x| <= ~1e4 keeps hi*2^-6 / lo*2^6 in f16 range)
  hipLaunchKernelGGL(dss::normalize_rows_split_kernel, dim3(blocks), dim3(256), 0, s, feats, hi, lo, (int)rows, D,
                     normalize ? eps : -1.0f);
  DSS_CHECK_LAUNCH("normalize_rows_split");
  const int ldw = dss_affinity_ld(N);
  const int nbk = (ldw / 64 + 1) / 2;
  const long nblocks = (long)(nbk * (nbk + 1) / 2) * B;
  DSS_REQUIRE(nblocks < 2147483647L, "dss_affinity_split: too many blocks (%ld)", nblocks);
  if (w_u16)
    hipLaunchKernelGGL(dss::gram_split_kernel<uint16_t>, dim3((unsigned)nblocks), dim3(256), 0, s, hi, lo,
                       (uint16_t*)W, N, D, ldw, 1, dss_affinity_elems(N), B);
  else
    hipLaunchKernelGGL(dss::gram_split_kernel<float>, dim3((unsigned)nblocks), dim3(256), 0, s, hi, lo, (float*)W, N,
                       D, ldw, threshold_at_zero ? 1 : 0, dss_affinity_elems(N), B);
  DSS_CHECK_LAUNCH("gram_split");
  return DSS_OK;
}

}  // namespace dss

extern "C" int dss_affinity_split(const float* feats, float* W, int B, int N, int D, int normalize, float eps,
                                  int threshold_at_zero, void* workspace, size_t workspace_bytes, void* stream) {
  return dss::affinity_split(feats, W, 0, B, N, D, normalize, eps, threshold_at_zero, workspace, workspace_bytes,
                             stream);
}

extern "C" int dss_affinity_split_u16(const float* feats, uint16_t* W, int B, int N, int D, float eps,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return dss::affinity_split(feats, W, 1, B, N, D, 1, eps, 1, workspace, workspace_bytes, stream);
}

extern "C" int dss_affinity_fused_u16(const float* feats, uint16_t* W, int B, int N, int D, float eps, void* stream) {
  DSS_REQUIRE(feats && W, "dss_affinity_fused_u16: null pointer");
  DSS_REQUIRE(B > 0 && N > 0 && D > 0, "dss_affinity_fused_u16: bad shape B=%d N=%d D=%d", B, N, D);
  DSS_REQUIRE(D % dss::FK == 0, "dss_affinity_fused_u16: feature dim must be a multiple of %d (got %d)", dss::FK, D);
  const int ldw = dss_affinity_ld(N);
  const int nbk = (ldw / 64 + 1) / 2;
  const long nblocks = (long)(nbk * (nbk + 1) / 2) * B;
  DSS_REQUIRE(nblocks < 2147483647L, "dss_affinity_fused_u16: too many blocks (%ld)", nblocks);
  hipLaunchKernelGGL(dss::gram_f16_fused_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, feats, W, N,
                     D, ldw, eps, dss_affinity_elems(N), B);
  DSS_CHECK_LAUNCH("gram_f16_fused");
  return DSS_OK;
}

extern "C" size_t dss_affinity_split_workspace_bytes(int B, int N, int D) {
  if (B <= 0 || N <= 0 || D <= 0) return 0;
  return (size_t)2 * B * N * D * sizeof(dss::f16);
}
