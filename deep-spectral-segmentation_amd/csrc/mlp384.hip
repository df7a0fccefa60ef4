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
//     both operands agree.  fc2's weight is therefore pre-permuted once (dss_mlp_k384_pack) into exactly that order,
//     fragment-major (and fc1's into its own fragment order), so a chunk is a contiguous 24 KB LDS-DMA copy; 24 more MFMAs (12 column tiles x 2
//     k-steps) accumulate the chunk's contribution to all 384 outputs.
//   * the GELU of chunk c is cut into 15 stages (gelu_stage) placed between the fc1 MFMAs of chunk c+1 and the first
//     fc2 MFMAs of chunk c (independent work; double-buffered accumulators), the order pinned with scheduling groups.
//   * W1 / packed-W2 chunks (24 KB each) are double-buffered in LDS by LDS-DMA, fragment-major images like
//     linear384.hip; W1 runs one chunk ahead of W2.  One workgroup barrier per chunk.
//   * epilogue: + fc2 bias, f16, transpose through LDS, full 128-byte non-temporal lines (row-major or DSS_PLANAR64).
// HBM traffic per token row: 768 B in + 768 B out instead of 7.7 KB.
//
// STATUS (round 1): parity-green, 0.67-0.72 PFLOP/s = 860-920 us per 290-image forward against 815 us for the
// unfused pair (K-resident fc1+GELU kernel + library fc2) in the same process, so it is OPT-IN (DSS_MLP_FUSED=1).
// What was measured on the way (scripts/debug/mlp_ab.py): without the in-loop weight staging the kernel takes 737 us -
// every 128-row workgroup re-streams all 2.36 MB of weights from L2 (5 TB/s aggregate at this speed), which is why both
// weights are packed fragment-major (linear 1 KB DMA instructions: 946 -> 859 us); a third LDS buffer (requests two
// chunks ahead) spilled registers (1455 us); hipcc sinks the fragment reads next to their MFMAs unless the order is
// pinned with sched_group_barrier, and places the GELU's VALU work after, not between, the MFMAs it is independent
// of - the matrix pipe is ~45 % busy.  The data path is right; what is left is an asm-level software pipeline and
// more token rows per weight pass.
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
// k-step sp multiplies hidden unit unit(8 sp + e, hh); dss_mlp_k384_pack stores W2 in that order.

// The exact-erf GELU of kres.h (A&S 7.1.28) on four float2 pairs, cut into 15 stages so that the caller can place a
// stage between two MFMAs of an unrelated accumulator chain: stage J advances all four pairs by one step (4 independent
// v_pk_* instructions, or 8 v_and / v_rcp).  x: values in / GELU out; z, q: state.
static constexpr int GELU_STAGES = 15;
template <int J>
__device__ __forceinline__ void gelu_stage(f32x2 (&x)[4], f32x2 (&z)[4], f32x2 (&q)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (J == 0) {
      f32x2 ax;
      ax[0] = fabsf(x[j][0]);
      ax[1] = fabsf(x[j][1]);
      z[j] = ax * 0.70710678118654752f;
    } else if constexpr (J == 1) {
      q[j] = z[j] * 0.0000430638f + 0.0002765672f;
    } else if constexpr (J == 2) {
      q[j] = q[j] * z[j] + 0.0001520143f;
    } else if constexpr (J == 3) {
      q[j] = q[j] * z[j] + 0.0092705272f;
    } else if constexpr (J == 4) {
      q[j] = q[j] * z[j] + 0.0422820123f;
    } else if constexpr (J == 5) {
      q[j] = q[j] * z[j] + 0.0705230784f;
    } else if constexpr (J == 6) {
      q[j] = q[j] * z[j] + 1.0f;
    } else if constexpr (J >= 7 && J <= 10) {
      q[j] = q[j] * q[j];                                    // q^16
    } else if constexpr (J == 11) {
      q[j][0] = __builtin_amdgcn_rcpf(q[j][0]);
      q[j][1] = __builtin_amdgcn_rcpf(q[j][1]);
    } else if constexpr (J == 12) {
      z[j] = z[j] * 0.70710678118654752f;                    // |x| / 2
    } else if constexpr (J == 13) {
      q[j] = (1.0f - q[j]) * z[j];                           // (|x| / 2) erf
    } else if constexpr (J == 14) {
      x[j] = x[j] * 0.5f + q[j];
    }
  }
}

template <class T>
__global__ void mlp_pack_fc1_kernel(const T* __restrict__ W1, T* __restrict__ W1p) {
  // W1 [1536, 384] row-major -> [chunk 48][kstep 24][lane 64][e 8]: element = W1[32 c + li][16 s + 8 hh + e]
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= MK * MH) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, sk = (idx >> 9) % MKS, c = idx / (512 * MKS);
  const int li = lane & 31, hh = lane >> 5;
  W1p[idx] = W1[(size_t)(MHC * c + li) * MK + 16 * sk + 8 * hh + e];
}

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
  const unsigned g2 = (unsigned)((wave * NST) * 1024 + 16 * lane);   // both weights are packed fragment-major: linear
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
    for (int j = 0; j < NST; ++j) dma(src, g2 + 1024u * j, __builtin_amdgcn_readfirstlane(dst0 + 1024u * j));
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

  f32x16 out[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[t][r] = 0.f;

  // ---- one chunk of the pipeline, written out in issue order and fenced with sched_barrier(0) after every MFMA: with
  //      one wave per SIMD nothing else covers an LDS round trip or a dependent VALU chain, and left to itself the
  //      scheduler sinks every fragment read next to its MFMA (matrix pipe ~40 % busy).
  //        part 1: 24 + 1 MFMAs of fc1 for chunk `cn` (two chains, W1 fragments read 3 ahead) with the GELU of the
  //                first 8 hidden units of chunk `c` (-> fc2's k-step 0 operand) in the gaps;
  //        part 2: 24 MFMAs of fc2 for chunk `c` (k-step 0 of all 12 tiles, then k-step 1; fragments 3 ahead) with
  //                the GELU of the other 8 units (-> k-step 1 operand) in the gaps of the first 12.
  auto chunk = [&](int c, int cn, const f32x16& h0, const f32x16& h1, f32x16& n0, f32x16& n1) {
    constexpr int PF = 4;
    f32x2 xa[4], xb[4], z[4], q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xa[j][0] = h0[2 * j] + h1[2 * j];
      xa[j][1] = h0[2 * j + 1] + h1[2 * j + 1];
      xb[j][0] = h0[8 + 2 * j] + h1[8 + 2 * j];
      xb[j][1] = h0[8 + 2 * j + 1] + h1[8 + 2 * j + 1];
    }
    // ---------------------------------------------------------------- part 1
    const unsigned char* w1b = &Ws1[cn & 1][16 * lane];
    const float bcol = B1s[cn * MHC + li];
    V8 f[PF];
#pragma unroll
    for (int s = 0; s < PF - 1; ++s) f[s] = *reinterpret_cast<const V8*>(w1b + 1024 * s);
#pragma unroll
    for (int r = 0; r < 16; ++r) { n0[r] = 0.f; n1[r] = 0.f; }
    __builtin_amdgcn_sched_group_barrier(0x100, PF - 1, 0);   // pipeline fill: the first three W1 fragments
#define DSS_MLP_FC1_STEP(S)                                                                           \
    {                                                                                                 \
      if ((S) + PF - 1 < MKS) f[((S) + PF - 1) % PF] = *reinterpret_cast<const V8*>(w1b + 1024 * ((S) + PF - 1)); \
      if ((S) & 1) n1 = mfma32x32x16(f[(S) % PF], a[(S)], n1);                                        \
      else n0 = mfma32x32x16(f[(S) % PF], a[(S)], n0);                                                \
      if ((S) < GELU_STAGES) gelu_stage<((S) < GELU_STAGES ? (S) : 0)>(xa, z, q);                      \
      if ((S) + PF - 1 < MKS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
      if ((S) < GELU_STAGES) __builtin_amdgcn_sched_group_barrier(0x002, (S) == 0 ? 12 : ((S) == 11 ? 8 : 4), 0); \
    }
    DSS_MLP_FC1_STEP(0) DSS_MLP_FC1_STEP(1) DSS_MLP_FC1_STEP(2) DSS_MLP_FC1_STEP(3) DSS_MLP_FC1_STEP(4)
    DSS_MLP_FC1_STEP(5) DSS_MLP_FC1_STEP(6) DSS_MLP_FC1_STEP(7) DSS_MLP_FC1_STEP(8) DSS_MLP_FC1_STEP(9)
    DSS_MLP_FC1_STEP(10) DSS_MLP_FC1_STEP(11) DSS_MLP_FC1_STEP(12) DSS_MLP_FC1_STEP(13) DSS_MLP_FC1_STEP(14)
    DSS_MLP_FC1_STEP(15) DSS_MLP_FC1_STEP(16) DSS_MLP_FC1_STEP(17) DSS_MLP_FC1_STEP(18) DSS_MLP_FC1_STEP(19)
    DSS_MLP_FC1_STEP(20) DSS_MLP_FC1_STEP(21) DSS_MLP_FC1_STEP(22) DSS_MLP_FC1_STEP(23)
#undef DSS_MLP_FC1_STEP
    V8 p0, p1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { p0[2 * j] = from_f32<T>(xa[j][0]); p0[2 * j + 1] = from_f32<T>(xa[j][1]); }
    // ---------------------------------------------------------------- part 2 (its first reads + the bias step of fc1)
    const unsigned char* w2b = &Ws2[c & 1][16 * lane];
    auto frag = [&](int i) { return *reinterpret_cast<const V8*>(w2b + 2048 * (i % MT) + 1024 * (i / MT)); };
    V8 w[PF];
#pragma unroll
    for (int i = 0; i < PF - 1; ++i) w[i] = frag(i);
    {
      V8 fb, a_one;                                          // bias as a 25th k-step: (b1[unit], 0..) x (1, 0..)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        fb[e] = from_f32<T>((e == 0 && hh == 0) ? bcol : 0.0f);
        a_one[e] = from_f32<T>((e == 0 && hh == 0) ? 1.0f : 0.0f);
      }
      n1 = mfma32x32x16(fb, a_one, n1);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, PF - 1, 0);   // pipeline fill of fc2
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // the bias step of fc1
#define DSS_MLP_FC2_CORE(I)                                                                           \
      if ((I) + PF - 1 < 2 * MT) w[((I) + PF - 1) % PF] = frag((I) + PF - 1);                          \
      out[(I) % MT] = mfma32x32x16(w[(I) % PF], (I) < MT ? p0 : p1, out[(I) % MT]);
#define DSS_MLP_FC2_SCHED(I, NV)                                                                      \
      if ((I) + PF - 1 < 2 * MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
      if ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);
#define DSS_MLP_FC2_STEP_A(I, J0, J1)                                                                 \
    { DSS_MLP_FC2_CORE(I) gelu_stage<J0>(xb, z, q); gelu_stage<J1>(xb, z, q); DSS_MLP_FC2_SCHED(I, 16) }
#define DSS_MLP_FC2_STEP_B(I, J0)                                                                     \
    { DSS_MLP_FC2_CORE(I) gelu_stage<J0>(xb, z, q); DSS_MLP_FC2_SCHED(I, (J0) == 11 ? 8 : 4) }
#define DSS_MLP_FC2_STEP_C(I)                                                                         \
    {                                                                                                 \
      if ((I) == MT) {                                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
          p1[2 * j] = from_f32<T>(xb[j][0]); p1[2 * j + 1] = from_f32<T>(xb[j][1]);                   \
        }                                                                                             \
      }                                                                                               \
      DSS_MLP_FC2_CORE(I) DSS_MLP_FC2_SCHED(I, 0)                                                     \
    }
    // (the 15 GELU stages of the second half must run in order inside the first 12 steps: steps 0-2 take two each)
    DSS_MLP_FC2_STEP_A(0, 0, 1) DSS_MLP_FC2_STEP_A(1, 2, 3) DSS_MLP_FC2_STEP_A(2, 4, 5) DSS_MLP_FC2_STEP_B(3, 6)
    DSS_MLP_FC2_STEP_B(4, 7) DSS_MLP_FC2_STEP_B(5, 8) DSS_MLP_FC2_STEP_B(6, 9) DSS_MLP_FC2_STEP_B(7, 10)
    DSS_MLP_FC2_STEP_B(8, 11) DSS_MLP_FC2_STEP_B(9, 12) DSS_MLP_FC2_STEP_B(10, 13) DSS_MLP_FC2_STEP_B(11, 14)
    DSS_MLP_FC2_STEP_C(12) DSS_MLP_FC2_STEP_C(13) DSS_MLP_FC2_STEP_C(14) DSS_MLP_FC2_STEP_C(15) DSS_MLP_FC2_STEP_C(16)
    DSS_MLP_FC2_STEP_C(17) DSS_MLP_FC2_STEP_C(18) DSS_MLP_FC2_STEP_C(19) DSS_MLP_FC2_STEP_C(20) DSS_MLP_FC2_STEP_C(21)
    DSS_MLP_FC2_STEP_C(22) DSS_MLP_FC2_STEP_C(23)
  };

  // ---- pipeline.  Iteration c holds: W1 chunk c+1 and W2 chunk c in LDS, pre-activations of chunk c in registers;
  //      W1 chunk c+2 and W2 chunk c+1 are DMA'd into the buffers last read in iteration c-1.  The last iteration
  //      recomputes fc1 of chunk 47 into a dead accumulator (one basic block for every iteration).
  f32x16 ha0, ha1, hb0, hb1;
  stage_w1(0);
  stage_w2(0);
  stage_w1(1);
  wait_vm();
  __syncthreads();
  {
    f32x16 d0, d1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
    const unsigned char* w1b = &Ws1[0][16 * lane];           // fc1 of chunk 0 (nothing to overlap with yet)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ha0[r] = 0.f; ha1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < MKS; ++s) {
      const V8 f = *reinterpret_cast<const V8*>(w1b + 1024 * s);
      if (s & 1) ha1 = mfma32x32x16(f, a[s], ha1);
      else ha0 = mfma32x32x16(f, a[s], ha0);
    }
    V8 fb, a_one;
    const float bcol = B1s[li];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      fb[e] = from_f32<T>((e == 0 && hh == 0) ? bcol : 0.0f);
      a_one[e] = from_f32<T>((e == 0 && hh == 0) ? 1.0f : 0.0f);
    }
    ha1 = mfma32x32x16(fb, a_one, ha1);
  }
  for (int c = 0; c < MNCH; c += 2) {
    if (c + 2 < MNCH) stage_w1(c + 2);
    if (c + 1 < MNCH) stage_w2(c + 1);
    chunk(c, c + 1, ha0, ha1, hb0, hb1);                      // c + 1 <= 47 always (48 chunks, c even)
    wait_vm();
    __syncthreads();
    if (c + 3 < MNCH) stage_w1(c + 3);
    if (c + 2 < MNCH) stage_w2(c + 2);
    chunk(c + 1, c + 2 < MNCH ? c + 2 : MNCH - 1, hb0, hb1, ha0, ha1);
    wait_vm();
    __syncthreads();
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

extern "C" int dss_mlp_k384_pack(const void* W1, const void* W2, void* W1_packed, void* W2_packed, int dtype,
                                 void* stream) {
  DSS_REQUIRE(W1 && W2 && W1_packed && W2_packed, "dss_mlp_k384_pack: null pointer");
  const int n = dss::MK * dss::MH;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case DSS_F16:
      hipLaunchKernelGGL((dss::mlp_pack_fc1_kernel<dss::f16>), dim3(n / 256), dim3(256), 0, s, (const dss::f16*)W1,
                         (dss::f16*)W1_packed);
      hipLaunchKernelGGL((dss::mlp_pack_fc2_kernel<dss::f16>), dim3(n / 256), dim3(256), 0, s, (const dss::f16*)W2,
                         (dss::f16*)W2_packed);
      break;
    case DSS_BF16:
      hipLaunchKernelGGL((dss::mlp_pack_fc1_kernel<dss::bf16>), dim3(n / 256), dim3(256), 0, s, (const dss::bf16*)W1,
                         (dss::bf16*)W1_packed);
      hipLaunchKernelGGL((dss::mlp_pack_fc2_kernel<dss::bf16>), dim3(n / 256), dim3(256), 0, s, (const dss::bf16*)W2,
                         (dss::bf16*)W2_packed);
      break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_mlp_k384_pack: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  }
  DSS_CHECK_LAUNCH("mlp_pack");
  return DSS_OK;
}

extern "C" int dss_mlp_k384(const void* A, const void* W1_packed, const void* b1, const void* W2_packed,
                            const void* b2, void* C, int M, int out_layout, int dtype, void* stream) {
  const void* W1 = W1_packed;
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
