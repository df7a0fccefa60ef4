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
  const WsymLayout LW = wsym_layout(N);             // where tile (ti, tj) lives: a full tile, or mini tiles of the edge strip
  auto store_tile = [&](const f32x16& acc, int rsub, int csub, bool computed) {
    const int col = cj0 + csub + li;
    if (!wsym_has(LW, tj, csub + li)) return;        // a column of the edge the strip does not keep (>= N: zero anyway)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = rsub + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const int row = ri0 + lr;
      float v = computed ? acc[r] : 0.f;
      if (relu) v = fmaxf(v, 0.f);
      if (row >= N || col >= N) v = 0.f;
      WE* dst = Wb + wsym_at(LW, ti, tj, lr, csub + li);
      if constexpr (sizeof(WE) == 4) {
        *dst = v;
      } else {
        const float q = __builtin_rintf(fminf(fmaxf(v, 0.f), 1.0f) * 65535.0f);
        *dst = (WE)(unsigned)q;
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
  const WsymLayout LW = wsym_layout(N);             // where tile (ti, tj) lives: a full tile, or mini tiles of the edge strip
  auto store_tile = [&](const f32x16& acc, int rsub, int csub, bool computed) {
    const int col = cj0 + csub + li;
    if (!wsym_has(LW, tj, csub + li)) return;        // a column of the edge the strip does not keep (>= N: zero anyway)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = rsub + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const int row = ri0 + lr;
      float v = computed ? acc[r] : 0.f;
      if (relu) v = fmaxf(v, 0.f);
      if (row >= N || col >= N) v = 0.f;
      WE* dst = Wb + wsym_at(LW, ti, tj, lr, csub + li);
      if constexpr (sizeof(WE) == 4) {
        *dst = v;
      } else {
        const float q = __builtin_rintf(fminf(fmaxf(v, 0.f), 1.0f) * 65535.0f);
        *dst = (WE)(unsigned)q;
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
  const WsymLayout LW = wsym_layout(N);             // where tile (ti, tj) lives: a full tile, or mini tiles of the edge strip
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
      if (wsym_has(LW, tj, lc)) *reinterpret_cast<u32x2*>(Wb + wsym_at(LW, ti, tj, lrow, lc)) = out;   // (4 columns: one mini-tile row)
    }
  };
  store_tile(acc00, 0, 0);
  store_tile(acc01, 0, 32);
  store_tile(acc10, 32, 0);
  store_tile(acc11, 32, 32);
}


// ================================================================================================
// The pipeline's affinity build (round 3): features that come straight out of the ViT are handed over in f16 together
// with their inverse norms (kfeatures_finalize_kernel below - the K projection's bias add, the CLS drop, the fp32
// features the caller gets back, the f16 copy and 1 / |x| in ONE pass instead of two torch passes), and the Gram kernel
// works from those:
//   * why: gram_f16_fused (fp32 in, 128 x 128 tiles) was bound by the L2 -> CU path, not by HBM - every block pulled
//     2 x 128 rows x D fp32 = 393 KB through the vector memory path for 16 K outputs, 28.7 GB per 2030-image launch
//     against 4.45 GB of HBM traffic, converting while staging (cvt + dot2 + ds_write per element, two barriers per 32
//     columns).  f16 rows halve those bytes, 256 x 128 tiles take another quarter off (10.8 GB), and the panels arrive by
//     LDS-DMA: no staging registers, no conversion, one barrier per 32-column stage with two stages of DMA in flight.
//   * workgroup = 8 waves (4 x 2) on a 256-row x 128-column tile, wave = 64 x 64 = one packed storage tile (64
//     accumulator registers, 92 VGPRs: two workgroups per CU); transposed product as before (a lane owns rows of W, its
//     registers run along the columns: 8-byte stores into the storage tile).
//   * measured (2030 images, N = 900, D = 384): 2.43 ms for gram_f16_fused -> 1.51 ms.  What bounds it now is the request
//     rate of the L2 -> LDS path, not HBM and not the MFMAs: 40 600 blocks x 12 stages x 24 KB = 11.7 GB in 64-byte row
//     segments = 7.7 TB/s (~120 G requests/s; the fp32 kernel's full-line loads moved 12 TB/s at ~94 G/s).  A 256 x 256
//     tile with 128 accumulator registers per wave (one workgroup per CU, 7.2 GB) measured the same 1.58 ms: fewer
//     bytes, but nothing to run under a block's prologue and output stores.
//   * the upper block triangle in these tiles: row block bi (storage-tile rows 4 bi ..) needs column blocks cj >= 2 bi.
//   * LDS stage = [row panel 256 x 64 B][column panel 128 x 64 B] = 24 KB, a ring of three; one DMA piece = 16 rows x
//     64 B, 16-byte chunk c of row r at position c ^ ((r >> 2) & 3) (swizzle on the SOURCE address): the 16 lanes of a
//     ds_read_b128 group (16 rows, one chunk) hit the 16 slots of a bank row once.
//   * epilogue: w = <x_i, x_j> r_i r_j with the inverse norms staged in LDS, relu + round(65535 w) by v_cvt_pknorm_u16,
//     rows / columns >= N written as 0.
// Algorithmic bytes per image: 2 N D (f16 features) + 4 N (norms) + N (N + 1) (the packed 16-bit upper triangle).
static constexpr int G3R = 256;      // block tile rows
static constexpr int G3C = 128;      // block tile columns
static constexpr int G3K = 32;       // feature columns per stage (64-byte rows)
static constexpr long G3PASS_BYTES = 2883584;   // 2.75 MB of column panels per group (+ ~1.2 MB of row panels in flight < 4 MB L2)

__global__ __launch_bounds__(512, 4) void gram_f16_dma_kernel(const f16* __restrict__ feats, const float* __restrict__ rnorm,
                                                              uint16_t* __restrict__ W, int N, int D, int ldw,
                                                              size_t w_stride, int nimg, int nblk) {
  constexpr int RING = 3;                                // stage buffers: two stages' DMA in flight while one is consumed
  constexpr int RP = G3R * 64, CP = G3C * 64;            // bytes per row / column panel per stage
  __shared__ __attribute__((aligned(1024))) unsigned char lds[RING][RP + CP];
  __shared__ __attribute__((aligned(16))) float rn[G3R + G3C];   // inverse norms of the block's rows, then columns (0 past N)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int nt = ldw / 64, ncj = (nt + 1) / 2;
  int img, rem;
  {  // XCD-aware order: all blocks of one image share an XCD
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
  // Block order inside an image.  Row block bi (256 rows) holds column blocks 2 bi .. ncj - 1 (128 columns).  The blocks of an
  // image run on ONE XCD, ~64 at a time: in plain row-major order those touch every column panel of the image again for
  // every row block, and once an image's f16 features outgrow the XCD's 4 MB L2 (dino_vitb8 at 480 x 480: 5.5 MB) the column
  // panels come back through the fabric each time - 19.8 GB moved for 9.5 GB algorithmic per 512 images (round 3).  So the
  // columns are cut into `npass` groups of <= ~2.75 MB of panels, the row blocks of the upper triangle run per group, and a
  // group's column panels stay L2-resident under all of them (row panels are read once per group they meet).
  const int npass = (int)(((long)N * D * 2 + G3PASS_BYTES - 1) / G3PASS_BYTES);
  int bi = 0, cj = 0;
  if (npass <= 1) {
    while (rem >= ncj - 2 * bi) { rem -= ncj - 2 * bi; ++bi; }
    cj = 2 * bi + rem;
  } else {
    const int per = (ncj + npass - 1) / npass;             // column blocks per group
    for (int c0 = 0; c0 < ncj; c0 += per) {
      const int c1 = min(c0 + per, ncj);
      bool found = false;
      for (bi = 0; 2 * bi < c1; ++bi) {
        const int lo = max(2 * bi, c0), cnt = c1 - lo;
        if (rem < cnt) { cj = lo + rem; found = true; break; }
        rem -= cnt;
      }
      if (found) break;
    }
  }
  const int I0 = bi * G3R, J0 = cj * G3C;
  const f16* F = feats + (long)img * N * D;
  const float* R = rnorm + (long)img * N;
  uint16_t* Wb = W + img * w_stride;
  const int ti = 4 * bi + wr, tj = 2 * cj + wc;          // this wave's storage tile
  const bool compute = ti < nt && tj < nt && tj >= ti;

  auto uniform_ptr = [](const void* p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
  };
  const unsigned char* fsrc = uniform_ptr(F);
  typedef __attribute__((address_space(3))) void* lds3_t;
  const unsigned lds0 = (unsigned)(size_t)(lds3_t)(&lds[0][0]);
  auto dma16 = [&](unsigned off, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(dst), "v"(off), "s"(fsrc) : "memory");
  };
  // a stage = 16 + 8 pieces of 16 rows; wave w moves row pieces 2 w, 2 w + 1 and column piece w: 3 DMA instructions
  const unsigned rowbytes = (unsigned)D * 2u;
  auto issue = [&](int s) {
    const unsigned sbase = lds0 + (unsigned)((s % RING) * (RP + CP));
    const unsigned scol = (unsigned)(s * 64);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int piece = j < 2 ? 2 * wave + j : wave;      // piece of the row panel (j < 2) / of the column panel
      const int r = 16 * piece + (lane >> 2);
      const unsigned c = (unsigned)((lane & 3) ^ ((r >> 2) & 3));
      int g = (j < 2 ? I0 : J0) + r;
      g = g < N ? g : N - 1;                              // rows past N: finite data, zeroed on output
      const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + (unsigned)((j < 2 ? 0 : RP) + piece * 1024));
      dma16((unsigned)g * rowbytes + scol + 16u * c, dst);
    }
  };
  const int ns = D / G3K;
#pragma unroll
  for (int s = 0; s < RING - 1; ++s)
    if (s < ns) issue(s);
  if (tid < G3R + G3C) {   // visible to everyone behind the first stage barrier
    const int g = tid < G3R ? I0 + tid : J0 + tid - G3R;
    rn[tid] = g < N ? R[g] : 0.f;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment byte offsets inside a panel: row (base + li), chunk 2 kk + hh at position ^ ((row >> 2) & 3)
  // (row bases are multiples of 32, so the swizzle term depends on li only)
  const unsigned sw = (unsigned)((li >> 2) & 3);
  unsigned foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = (unsigned)(li * 64) + ((((unsigned)(2 * kk + hh)) ^ sw) << 4);

  // One stage (8 MFMAs per wave) is far too short to cover an L2 / HBM round trip, so the DMA runs two stages ahead and
  // the wait is counted: the younger stage's three pieces per wave stay in flight.
  for (int s = 0; s < ns; ++s) {
    if (s + 1 < ns) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // stage s complete for everyone; buffer (s - 1) % RING is free
    asm volatile("" ::: "memory");
    if (s + RING - 1 < ns) issue(s + RING - 1);
    if (compute) {
      const unsigned char* rp = &lds[s % RING][0] + (wr * 64) * 64;
      const unsigned char* cp = &lds[s % RING][RP] + (wc * 64) * 64;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const f16x8 r0 = *reinterpret_cast<const f16x8*>(rp + foff[kk]);                // row panel -> "B" operand
        const f16x8 r1 = *reinterpret_cast<const f16x8*>(rp + 32 * 64 + foff[kk]);
        const f16x8 c0 = *reinterpret_cast<const f16x8*>(cp + foff[kk]);                // column panel -> "A" operand
        const f16x8 c1 = *reinterpret_cast<const f16x8*>(cp + 32 * 64 + foff[kk]);
        acc[0][0] = mfma32x32x16(c0, r0, acc[0][0]);
        acc[0][1] = mfma32x32x16(c1, r0, acc[0][1]);
        acc[1][0] = mfma32x32x16(c0, r1, acc[1][0]);
        acc[1][1] = mfma32x32x16(c1, r1, acc[1][1]);
      }
    }
  }
  if (!compute) return;
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const WsymLayout LW = wsym_layout(N);             // where tile (ti, tj) lives: a full tile, or mini tiles of the edge strip
  const bool edge = tj >= LW.ntf;                   // wave-uniform
  uint16_t* tile = Wb + wsym_at(LW, ti, tj, 0, 0);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int lrow = 32 * a + li;                         // this lane's row inside the tile
    const float rr = rn[wr * 64 + lrow];
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
      const f32x16& A = acc[a][b2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int lc = 32 * b2 + 8 * g + 4 * hh;          // 4 consecutive columns of the tile
        const f32x4 rc = *reinterpret_cast<const f32x4*>(&rn[G3R + wc * 64 + lc]);
        const u16x2 q0 = __builtin_amdgcn_cvt_pknorm_u16(A[4 * g] * rr * rc[0], A[4 * g + 1] * rr * rc[1]);
        const u16x2 q1 = __builtin_amdgcn_cvt_pknorm_u16(A[4 * g + 2] * rr * rc[2], A[4 * g + 3] * rr * rc[3]);
        const u32x2 out = {__builtin_bit_cast(unsigned, q0), __builtin_bit_cast(unsigned, q1)};
        if (!edge) *reinterpret_cast<u32x2*>(tile + lrow * 64 + lc) = out;
        else if (lc < 4 * LW.e4) *reinterpret_cast<u32x2*>(tile + ((lc >> 2) * 64 + lrow) * 4) = out;   // mini tile lc / 4, row lrow
      }
    }
  }
}

// K projection output -> what the caller and the affinity build need, in one pass over the rows:
//   kproj [B, T, D] f32 (the GEMM's raw output, token 0 = CLS) (+ bias [D])
//   -> k32 [B, T-1, D] f32 (extract.py:96-98: CLS dropped), k16 the same rounded to f16, rnorm [B, T-1] = 1 / max(|x16|, eps)
// (the norm of the ROUNDED row, so that w_ii = 1 exactly).  One wave per output row.
__global__ __launch_bounds__(256) void kfeatures_finalize_kernel(const float* __restrict__ kproj, const float* __restrict__ bias,
                                                                 float* __restrict__ k32, f16* __restrict__ k16,
                                                                 float* __restrict__ rnorm, long rows_out, int T, int D,
                                                                 float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int N = T - 1;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows_out; row += (long)gridDim.x * 4) {
    const long b = row / N, n = row - b * N;
    const float* src = kproj + (b * T + n + 1) * (long)D;
    float* d32 = k32 + row * (long)D;
    f16* d16 = k16 + row * (long)D;
    float ss = 0.f;
    for (int c = lane; c < (D >> 2); c += 64) {
      f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * c);
      if (bias) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + 4 * c);
        v += bb;
      }
      *reinterpret_cast<f32x4*>(d32 + 4 * c) = v;
      f16x4 h;
#pragma unroll
      for (int i = 0; i < 4; ++i) { h[i] = (f16)v[i]; const float r = (float)h[i]; ss += r * r; }
      *reinterpret_cast<f16x4*>(d16 + 4 * c) = h;
    }
    ss = wave_sum(ss);
    if (lane == 0) rnorm[row] = 1.0f / fmaxf(sqrtf(ss), eps);
  }
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
extern "C" size_t dss_affinity_elems(int N) { return N > 0 ? dss::wsym_elems(N) : 0; }

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

extern "C" int dss_affinity_f16_u16(const void* feats16, const float* rnorm, uint16_t* W, int B, int N, int D, void* stream) {
  DSS_REQUIRE(feats16 && rnorm && W, "dss_affinity_f16_u16: null pointer");
  DSS_REQUIRE(B > 0 && N > 0 && D > 0, "dss_affinity_f16_u16: bad shape B=%d N=%d D=%d", B, N, D);
  DSS_REQUIRE(D % dss::G3K == 0, "dss_affinity_f16_u16: feature dim must be a multiple of %d (got %d)", dss::G3K, D);
  DSS_REQUIRE((long)N * D * 2 < 4294967296L, "dss_affinity_f16_u16: one image's features exceed the DMA's 32-bit offsets");
  const int ldw = dss_affinity_ld(N);
  const int nt = ldw / 64, ncj = (nt + 1) / 2;
  int nblk = 0;                                           // row block bi (256 rows) x column blocks 2 bi .. ncj - 1 (128 columns)
  for (int bi = 0; 2 * bi < ncj; ++bi) nblk += ncj - 2 * bi;
  const long nblocks = (long)nblk * B;
  DSS_REQUIRE(nblocks < 2147483647L, "dss_affinity_f16_u16: too many blocks (%ld)", nblocks);
  hipLaunchKernelGGL(dss::gram_f16_dma_kernel, dim3((unsigned)nblocks), dim3(512), 0, (hipStream_t)stream,
                     (const dss::f16*)feats16, rnorm, W, N, D, ldw, dss_affinity_elems(N), B, nblk);
  DSS_CHECK_LAUNCH("gram_f16_dma");
  return DSS_OK;
}

extern "C" int dss_kfeatures_finalize(const float* kproj, const float* bias, float* k32, void* k16, float* rnorm, int B,
                                      int T, int D, float eps, void* stream) {
  DSS_REQUIRE(kproj && k32 && k16 && rnorm, "dss_kfeatures_finalize: null pointer");
  DSS_REQUIRE(B > 0 && T > 1 && D > 0 && D % 4 == 0, "dss_kfeatures_finalize: bad shape B=%d T=%d D=%d", B, T, D);
  const long rows = (long)B * (T - 1);
  long blocks = (rows + 3) / 4;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(dss::kfeatures_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, kproj, bias,
                     k32, (dss::f16*)k16, rnorm, rows, T, D, eps);
  DSS_CHECK_LAUNCH("kfeatures_finalize");
  return DSS_OK;
}

extern "C" size_t dss_affinity_split_workspace_bytes(int B, int N, int D) {
  if (B <= 0 || N <= 0 || D <= 0) return 0;
  return (size_t)2 * B * N * D * sizeof(dss::f16);
}
