// mlp384.hip - DINO's Mlp (fc1 -> GELU -> fc2) of the D = 384 models in ONE kernel; the 4x-wide hidden activations
// never leave the registers.
//
// Replaces  x = self.fc2(self.act(self.fc1(x)))  of DINO's Mlp (reached from
// extract/extract.py:94).  Unfused, the pair writes and re-reads a [M, 1536] f16 tensor per layer (1.6 GB per
// 290-image forward: a third of the layer's HBM traffic) and runs the GELU as a VALU phase the matrix pipe waits
// for.  Here:
//   * a wave keeps its 32 token rows x 384 (A of fc1: 24 fragments = 96 VGPRs) AND its 32 x 384 fp32 output
//     accumulators (12 MFMA tiles = 192 registers) resident; one wave per SIMD (512 registers per lane).
//   * the hidden dimension streams in chunks of 32 units.  Per chunk: 24 (+1 bias) MFMAs of fc1 give the 32 x 32
//     pre-activations, the exact-erf GELU runs on them in registers, and - this is the point - the activated
//     values are ALREADY the B operand of fc2's MFMAs: fc1's accumulator layout (lane = token row, 16 registers =
//     16 hidden units) is a legal k-ordering of a 32x32x16 MFMA, because the k index of an MFMA is a dummy as long as
//     both operands agree.  fc2's weight is therefore pre-permuted on the host (dss_mlp_k384_pack_fc2) into exactly
//     that order, fragment-major, so its chunk is a contiguous 24 KB LDS-DMA copy; 24 more MFMAs (12 column tiles x 2
//     k-steps) accumulate the chunk's contribution to all 384 outputs.
//   * the GELU of chunk c is issued together with the fc1 MFMAs of chunk c+1 (independent; double-buffered
//     accumulators): VALU and matrix pipe overlap inside the one wave.
//   * W1 / packed-W2 chunks (24 KB each) are double-buffered in LDS by LDS-DMA, fragment-major images like
//     linear384.hip; W1 runs one chunk ahead of W2.  One workgroup barrier per chunk.
//   * epilogue: + fc2 bias, f16, transpose through LDS, full 128-byte non-temporal lines (row-major or DSS_PLANAR64).
// HBM traffic per token row: 768 B in + 768 B out instead of 7.7 KB.
//
// STATUS (round 1): parity-green, 0.66-0.69 PFLOP/s = 890-940 us per 290-image forward against 830 us for the
// unfused pair (K-resident fc1+GELU kernel + library fc2), so it is OPT-IN (DSS_MLP_FUSED=1).  With one wave per SIMD
// every LDS round trip and every dependent VALU chain must be covered by this wave's own instruction stream; hipcc
// sinks the fragment reads next to their MFMAs (sched_group_barrier pinning of both GEMMs was tried: slower), so the
// matrix pipe is ~40 % busy.  The data path is right; the remaining work is an asm-level software pipeline.
#include "common.h"
#include "kres.h"

namespace dss {

static constexpr int MK = 384;             // embedding width
static constexpr int MKS = MK / 16;        // 24 k-steps of fc1
static constexpr int MH = 1536;            // hidden width
static constexpr int MHC = 32;             // hidden units per chunk
static constexpr int MNCH = MH / MHC;      // 48 chunks
static constexpr int MT = MK / 32;         // 12 output column tiles of fc2
static constexpr int MWAVES = 4;
static constexpr int MTHREADS = 64 * MWAVES;
static constexpr int MROWS = 32 * MWAVES;  // token rows per workgroup
static constexpr int MW_BYTES = MHC * MK * 2;   // 24576: one chunk of W1 (32 x 384) or of packed W2 (384 x 32)

// hidden unit (inside a chunk) held by register r of fc1's accumulator in the half-wave hh:
//   unit(r, hh) = (r & 3) + 8 (r >> 2) + 4 hh          (the C/D layout of v_mfma_f32_32x32x16)
// fc2 consumes registers 0..7 as its k-step 0 and 8..15 as its k-step 1, slot e = r & 7, so the k-slot (hh, e) of
// k-step sp multiplies hidden unit unit(8 sp + e, hh); dss_mlp_k384_pack_fc2 stores W2 in that order.

template <class T>
__global__ void mlp_pack_fc2_kernel(const T* __restrict__ W2, T* __restrict__ W2p) {
  // W2 [384, 1536] row-major -> [chunk 48][tile 12][kstep 2][lane 64][e 8]
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one output element
  if (idx >= MK * MH) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, sp = (idx >> 9) & 1, t = (idx >> 10) % MT, c = idx / (1024 * MT);
  const int li = lane & 31, hh = lane >> 5, r = 8 * sp + e;
  const int unit = (r & 3) + 8 * (r >> 2) + 4 * hh;
  W2p[idx] = W2[(size_t)(32 * t + li) * MH + MHC * c + unit];
}

template <class T>
__global__ __launch_bounds__(MTHREADS, 1) void mlp_k384_kernel(const T* __restrict__ A, const T* __restrict__ W1,
                                                              const T* __restrict__ b1, const T* __restrict__ W2p,
                                                              const T* __restrict__ b2, T* __restrict__ C, int M,
                                                              int planar) {
  typedef typename vec8<T>::type V8;
  typedef typename vec4<T>::type V4;
  __shared__ __attribute__((aligned(256))) unsigned char Ws1[2][MW_BYTES];
  __shared__ __attribute__((aligned(256))) unsigned char Ws2[2][MW_BYTES];
  __shared__ __attribute__((aligned(16))) float B1s[MH];
  __shared__ __attribute__((aligned(16))) float B2s[MK];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int mrem = M - blockIdx.x * MROWS;                   // rows of this workgroup that exist (> 0)
  const int rloc = wave * 32;
  const bool block_full = mrem >= MROWS;

  // ---- this lane's token row as the B-operand fragments of fc1: k = 16 s + 8 hh + e -----------------------------
  V8 a[MKS];
  {
    const long r = (long)blockIdx.x * MROWS + min(rloc + li, mrem - 1);
#pragma unroll
    for (int s = 0; s < MKS; ++s) a[s] = *reinterpret_cast<const V8*>(A + r * MK + 16 * s + 8 * hh);
  }

  // ---- chunk staging (LDS-DMA, 1 KB per wave instruction, 6 + 6 per wave per chunk) ------------------------------
  constexpr int NST = MKS / MWAVES;                          // 6
  const unsigned g1 = (unsigned)(li * (MK * 2) + 16 * hh + 32 * (wave * NST));   // W1: gather, fragment-major image
  const unsigned g2 = (unsigned)((wave * NST) * 1024 + 16 * lane);               // packed W2: already fragment-major
  auto dma = [&](const unsigned char* src, unsigned off, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(dst), "v"(off), "s"(src) : "memory");
  };
  auto stage_w1 = [&](int c) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(W1) + (size_t)c * MW_BYTES;
    const unsigned dst0 = (unsigned)(size_t)(lds_ptr_t)(&Ws1[c & 1][wave * NST * 1024]);
#pragma unroll
    for (int j = 0; j < NST; ++j) dma(src, g1 + 32u * j, __builtin_amdgcn_readfirstlane(dst0 + 1024u * j));
  };
  auto stage_w2 = [&](int c) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(W2p) + (size_t)c * MW_BYTES;
    const unsigned dst0 = (unsigned)(size_t)(lds_ptr_t)(&Ws2[c & 1][wave * NST * 1024]);
#pragma unroll
    for (int j = 0; j < NST; ++j) dma(src, g2 + 1024u * j, __builtin_amdgcn_readfirstlane(dst0 + 1024u * j));
  };
  auto wait_vm = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  for (int i = tid; i < MH; i += MTHREADS) B1s[i] = to_f32<T>(b1[i]);
  for (int i = tid; i < MK; i += MTHREADS) B2s[i] = to_f32<T>(b2[i]);

  // ---- fc1 of chunk c: pre-activations = W1_chunk . A^T + b1 in two MFMA chains (even / odd k-steps) ------------
  auto fc1 = [&](int c, f32x16& h0, f32x16& h1) {
    const unsigned char* wb = &Ws1[c & 1][16 * lane];
    const float bcol = B1s[c * MHC + li];
    V8 f[3];
    f[0] = *reinterpret_cast<const V8*>(wb);
    f[1] = *reinterpret_cast<const V8*>(wb + 1024);
#pragma unroll
    for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < MKS; ++s) {
      if (s + 2 < MKS) f[(s + 2) % 3] = *reinterpret_cast<const V8*>(wb + 1024 * (s + 2));
      if (s & 1) h1 = mfma32x32x16(f[s % 3], a[s], h1);     // D[unit][row] += W1[unit][k] * A[row][k]
      else h0 = mfma32x32x16(f[s % 3], a[s], h0);
    }
    V8 fb, a_one;                                            // bias as a 25th k-step: (b1[unit], 0..) x (1, 0..)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      fb[e] = from_f32<T>((e == 0 && hh == 0) ? bcol : 0.0f);
      a_one[e] = from_f32<T>((e == 0 && hh == 0) ? 1.0f : 0.0f);
    }
    h1 = mfma32x32x16(fb, a_one, h1);
  };

  // ---- GELU of a chunk's pre-activations -> the two B-operand fragments of fc2 ---------------------------------
  auto activate = [&](const f32x16& h0, const f32x16& h1, V8& p0, V8& p1) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {                            // registers 8q .. 8q+7 = fc2 k-step q
      f32x2 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j][0] = h0[8 * q + 2 * j] + h1[8 * q + 2 * j];
        v[j][1] = h0[8 * q + 2 * j + 1] + h1[8 * q + 2 * j + 1];
      }
      gelu_erf2xn<4>(v);
      V8& p = q ? p1 : p0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { p[2 * j] = from_f32<T>(v[j][0]); p[2 * j + 1] = from_f32<T>(v[j][1]); }
    }
  };

  f32x16 out[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[t][r] = 0.f;

  // ---- fc2 contribution of one chunk: out[tile] += W2p[chunk][tile][kstep] . act^T --------------------------------
  auto fc2 = [&](int c, const V8& p0, const V8& p1) {
    const unsigned char* wb = &Ws2[c & 1][16 * lane];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const V8 w0 = *reinterpret_cast<const V8*>(wb + 2048 * t);
      const V8 w1 = *reinterpret_cast<const V8*>(wb + 2048 * t + 1024);
      out[t] = mfma32x32x16(w0, p0, out[t]);                 // D[outcol][row] += W2[outcol][unit] * act[row][unit]
      out[t] = mfma32x32x16(w1, p1, out[t]);
    }
  };

  // ---- pipeline.  Iteration c holds: W1 chunk c+1 and W2 chunk c in LDS, pre-activations of chunk c in registers.
  //      It activates chunk c while the matrix pipe runs fc1 of chunk c+1, then runs fc2 of chunk c; meanwhile W1
  //      chunk c+2 and W2 chunk c+1 are DMA'd into the buffers last read in iteration c-1.
  f32x16 ha0, ha1, hb0, hb1;
  stage_w1(0);
  stage_w2(0);
  stage_w1(1);
  wait_vm();
  __syncthreads();
  fc1(0, ha0, ha1);
  for (int c = 0; c < MNCH; c += 2) {
    // even chunk c: pre-activations in (ha0, ha1); compute chunk c+1 into (hb0, hb1)
    {
      if (c + 2 < MNCH) stage_w1(c + 2);
      if (c + 1 < MNCH) stage_w2(c + 1);
      V8 p0, p1;
      if (c + 1 < MNCH) fc1(c + 1, hb0, hb1);
      activate(ha0, ha1, p0, p1);
      fc2(c, p0, p1);
      wait_vm();
      __syncthreads();
    }
    // odd chunk c+1: pre-activations in (hb0, hb1); compute chunk c+2 into (ha0, ha1)
    {
      const int d = c + 1;
      if (d + 2 < MNCH) stage_w1(d + 2);
      if (d + 1 < MNCH) stage_w2(d + 1);
      V8 p0, p1;
      if (d + 1 < MNCH) fc1(d + 1, ha0, ha1);
      activate(hb0, hb1, p0, p1);
      fc2(d, p0, p1);
      wait_vm();
      __syncthreads();
    }
  }

  // ---- epilogue: + b2, f16, 64-column groups through a 4 KB transpose patch (the W buffers are free now) -----------
  unsigned char* stg = &Ws1[0][0] + wave * 4096;
  unsigned char* stg_w = stg + li * 128 + 8 * hh;            // writer: row li
  const unsigned stg_x = 16u * ((li >> 1) & 7);              // 16-byte slot p of row r lives at p ^ ((r >> 1) & 7)
  const int rq = lane >> 3, pq = lane & 7;                   // reader: row rq + 8 k, 16-byte piece pq
  const unsigned stg_ro = (unsigned)(rq * 128 + 16 * (pq ^ (rq >> 1)));
  const size_t ldc = planar ? 128 : (size_t)(MK * 2);
  const size_t gstride = planar ? (size_t)M * 128 : 128;
  unsigned char* cblk = reinterpret_cast<unsigned char*>(C) + (size_t)blockIdx.x * MROWS * ldc;
  const unsigned coff = (unsigned)((rloc + rq) * (unsigned)ldc + 16 * pq);
#pragma unroll
  for (int p = 0; p < MT / 2; ++p) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int t = 2 * p + half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(&B2s[32 * t + 8 * g + 4 * hh]);
        V4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = from_f32<T>(out[t][4 * g + i] + bv[i]);
        *reinterpret_cast<V4*>(stg_w + ((64u * half + 16 * g) ^ stg_x)) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // same-wave LDS write -> read (other lanes' data)
    unsigned char* cw = cblk + (size_t)p * gstride;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const V8 v = *reinterpret_cast<const V8*>(stg + (stg_ro ^ (64u * (k & 1))) + 1024 * k);
      if (block_full || rloc + rq + 8 * k < mrem)
        __builtin_nontemporal_store(v, reinterpret_cast<V8*>(cw + (size_t)(8 * k) * ldc + coff));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the patch is rewritten by the next group
  }
}

template <class T>
static int launch_mlp(const void* A, const void* W1, const void* b1, const void* W2p, const void* b2, void* C, int M,
                      int planar, hipStream_t s) {
  hipLaunchKernelGGL((mlp_k384_kernel<T>), dim3(ceil_div(M, MROWS)), dim3(MTHREADS), 0, s, (const T*)A, (const T*)W1,
                     (const T*)b1, (const T*)W2p, (const T*)b2, (T*)C, M, planar);
  return 0;
}

}  // namespace dss

extern "C" int dss_mlp_k384_pack_fc2(const void* W2, void* W2_packed, int dtype, void* stream) {
  DSS_REQUIRE(W2 && W2_packed, "dss_mlp_k384_pack_fc2: null pointer");
  const int n = dss::MK * dss::MH;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case DSS_F16:
      hipLaunchKernelGGL((dss::mlp_pack_fc2_kernel<dss::f16>), dim3(n / 256), dim3(256), 0, s, (const dss::f16*)W2,
                         (dss::f16*)W2_packed);
      break;
    case DSS_BF16:
      hipLaunchKernelGGL((dss::mlp_pack_fc2_kernel<dss::bf16>), dim3(n / 256), dim3(256), 0, s, (const dss::bf16*)W2,
                         (dss::bf16*)W2_packed);
      break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_mlp_k384_pack_fc2: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  }
  DSS_CHECK_LAUNCH("mlp_pack_fc2");
  return DSS_OK;
}

extern "C" int dss_mlp_k384(const void* A, const void* W1, const void* b1, const void* W2_packed, const void* b2,
                            void* C, int M, int out_layout, int dtype, void* stream) {
  DSS_REQUIRE(A && W1 && b1 && W2_packed && b2 && C, "dss_mlp_k384: null pointer");
  DSS_REQUIRE(M > 0, "dss_mlp_k384: need M > 0 (M=%d)", M);
  DSS_REQUIRE(out_layout == DSS_ROW_MAJOR || out_layout == DSS_PLANAR64,
              "dss_mlp_k384: out_layout must be DSS_ROW_MAJOR or DSS_PLANAR64 (got %d)", out_layout);
  hipStream_t s = (hipStream_t)stream;
  const int planar = out_layout == DSS_PLANAR64;
  switch (dtype) {
    case DSS_F16: dss::launch_mlp<dss::f16>(A, W1, b1, W2_packed, b2, C, M, planar, s); break;
    case DSS_BF16: dss::launch_mlp<dss::bf16>(A, W1, b1, W2_packed, b2, C, M, planar, s); break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_mlp_k384: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  }
  DSS_CHECK_LAUNCH("mlp_k384");
  return DSS_OK;
}
