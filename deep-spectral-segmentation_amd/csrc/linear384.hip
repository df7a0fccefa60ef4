// linear384.hip - Linear layers of the D = 384 DINO ViTs (vits16 / vits8) whose reduction dimension is the
// embedding width: qkv (384 -> 1152), attn.proj (384 -> 384), mlp.fc1 (384 -> 1536, + exact GELU).
//
// Replaces torch.nn.Linear / F.gelu inside DINO's Block (reached from extract/extract.py:94).
// These GEMMs write 1.5 - 4x more bytes than they read (M = 230k token rows, K = 384): a library GEMM spends them in
// prologue/epilogue (hipBLASLt: 330 - 410 us for qkv, 1.3 - 1.6 TB/s of output) and the erf-GELU is a separate
// 1.4 GB elementwise pass.  Design, all of it measured on MI355X (scripts/debug/linear_ab.py):
//   * K is RESIDENT: a wave keeps its 64 token rows of A - all 384 columns, 48 MFMA fragments = 192 VGPRs - in
//     registers for the whole kernel; there is no K loop over memory.
//   * W [N, 384] (<= 1.2 MB, L2-resident) streams through LDS in chunks of 32 output columns (24 KB, double
//     buffered) by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).  The LDS image is
//     FRAGMENT-MAJOR - 16-byte piece 64 s + lane is exactly the operand lane `lane` needs at k-step s - so the
//     compute loop reads ds_read_b128 at 16 * lane + 1024 * s: conflict-free, one address register.
//   * 8 waves per workgroup in two groups that PING-PONG: while group X issues the 48 MFMAs of a chunk (alone on the
//     matrix pipe, fragments prefetched 2 k-steps ahead), group Y runs the epilogue of its previous chunk (VALU GELU,
//     LDS transpose, global stores), then they swap.  Epilogue and MFMA phases overlap by construction; two
//     independent workgroups per CU were measured to run in lockstep instead (matrix pipe idle in every epilogue).
//   * the product is D[col][row] = W_chunk . A^T, so a lane owns token rows; the wave transposes its 64 x 64 output
//     tile (two chunks) through a private 8 KB LDS patch and stores FULL 128-byte lines, 8 rows per instruction.
//     Measured for qkv: 8-byte pieces 404 us, 64-byte half lines 310 us, full lines 268 us (same MFMA loop).
//   * bias is the accumulators' initial value; GELU is exact-erf by Abramowitz-Stegun 7.1.28
//     (erf z = 1 - (1 + a1 z + .. + a6 z^6)^-16, |error| <= 3e-7: one v_rcp, no v_exp), written on float2 so the
//     polynomial runs on v_pk_fma_f32 / v_pk_mul_f32.
#include "common.h"

namespace dss {

static constexpr int LK = 384;            // reduction dimension (embedding width of vits*)
static constexpr int LKS = LK / 16;       // 24 MFMA k-steps
static constexpr int LBN = 32;            // output columns per W chunk (one MFMA column tile)
static constexpr int LWAVES = 8;          // two ping-pong groups of 4
static constexpr int LTHREADS = 64 * LWAVES;
static constexpr int LBM = 64 * LWAVES;   // token rows per workgroup (64 per wave)
static constexpr int LMAXN = 2048;        // bias staged in LDS (fp32)
static constexpr int LGELU_ILP = 2;       // float2 pairs advanced in lockstep by the GELU (4 spills registers)
static constexpr int LCHUNK_BYTES = LBN * LK * 2;   // 24576

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// GELU(x) = x * 0.5 * (1 + erf(x / sqrt 2)), NP float2 at a time, in place.  erf by A&S 7.1.28 on
// z = |x| / sqrt 2: erf z = 1 - q^-16, q = 1 + a1 z + ... + a6 z^6 (|error| <= 3e-7: one v_rcp, no v_exp).  Evaluated
// as x/2 + (|x|/2) (1 - q^-16): the negative branch cancels to -(|x|/2) q^-16 with absolute error ~6e-8 |x|, far
// below the f16 rounding of the output.  The chain of one value is 17 dependent VALU ops (the probe measures
// latency-, not issue-bound execution), so NP pairs are advanced in lockstep: every step below is NP independent
// v_pk_* instructions.
template <int NP>
__device__ __forceinline__ void gelu_erf2xn(f32x2* x) {
  f32x2 z[NP], q[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    f32x2 ax;
    ax[0] = fabsf(x[j][0]);
    ax[1] = fabsf(x[j][1]);
    z[j] = ax * 0.70710678118654752f;
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = z[j] * 0.0000430638f + 0.0002765672f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 0.0001520143f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 0.0092705272f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 0.0422820123f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 0.0705230784f;
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = q[j] * z[j] + 1.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {                            // q^16 (inf for |x| > ~30: 1/inf = 0, erf = 1)
#pragma unroll
    for (int j = 0; j < NP; ++j) q[j] = q[j] * q[j];
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    q[j][0] = __builtin_amdgcn_rcpf(q[j][0]);
    q[j][1] = __builtin_amdgcn_rcpf(q[j][1]);
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) q[j] = (1.0f - q[j]) * (z[j] * 0.70710678118654752f);   // (|x|/2) erf
#pragma unroll
  for (int j = 0; j < NP; ++j) x[j] = x[j] * 0.5f + q[j];
}

template <class T, bool GELU>
__global__ __launch_bounds__(LTHREADS, 1) void linear_k384_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                                 const T* __restrict__ bias, T* __restrict__ C,
                                                                 int M, int N, int planar) {
  typedef typename vec8<T>::type V8;
  typedef typename vec4<T>::type V4;
  __shared__ __attribute__((aligned(256))) unsigned char Ws[2][LCHUNK_BYTES];
  __shared__ __attribute__((aligned(256))) unsigned char Stg[LWAVES][8192];
  __shared__ __attribute__((aligned(16))) float Bs[LMAXN];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const bool group_x = wave < LWAVES / 2;
  const int mrem = M - blockIdx.x * LBM;                   // rows of this workgroup that exist (> 0)
  const int rloc = wave * 64;                              // this wave's first row inside the workgroup
  const bool block_full = mrem >= LBM;

  // ---- this lane's two token rows as MFMA B-operand fragments: k = 16 s + 8 hh + e ---------------------------
  V8 a0[LKS], a1[LKS];
  {
    const long r0 = (long)blockIdx.x * LBM + min(rloc + li, mrem - 1);
    const long r1 = (long)blockIdx.x * LBM + min(rloc + 32 + li, mrem - 1);
#pragma unroll
    for (int s = 0; s < LKS; ++s) {
      // plain loads: a 128-byte line of A is touched by 8 of these instructions (4 k-steps x 2 halves); with
      // non-temporal loads it is re-fetched from HBM each time (measured: qkv 255 -> 291 us)
      a0[s] = *reinterpret_cast<const V8*>(A + r0 * LK + 16 * s + 8 * hh);
      a1[s] = *reinterpret_cast<const V8*>(A + r1 * LK + 16 * s + 8 * hh);
    }
  }

  // ---- W chunk staging by LDS-DMA: instruction j of wave w stages k-step s = 3 w + j (64 lanes x 16 B = 1 KB):
  //      lane (li, hh) fetches W[chunk col li][16 s + 8 hh .. + 8] and the hardware writes it at base + 16 * lane.
  constexpr int NST = LKS / LWAVES;                        // 3
  const unsigned gsrc0 = (unsigned)(li * (LK * 2) + 16 * hh + 32 * (wave * NST));
  auto stage = [&](int c) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(W) + (size_t)c * LCHUNK_BYTES;  // uniform
    const unsigned dst0 = (unsigned)(size_t)(lds_ptr_t)(&Ws[c & 1][wave * NST * 1024]);
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const unsigned off = gsrc0 + 32u * j;
      const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + 1024u * j);
      // inline asm, not the builtin: the compiler's alias model would put s_waitcnt vmcnt(0) in front of the very
      // next ds_read and expose the whole L2 latency; the consumers sit behind wait_vm() + a barrier
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(dst), "v"(off), "s"(src) : "memory");
    }
  };
  auto wait_vm = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  // phase barrier WITHOUT the release fence of __syncthreads(): that fence makes the compiler drain vmcnt to 0 (the
  // tile stores!) in front of every barrier.  Nothing crosses waves through memory inside the loop except the W
  // chunks, whose arrival is awaited explicitly (wait_dma) by the issuing waves.
  auto phase_barrier = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // vmcnt retires in issue order (gfx9: loads, LDS-DMA and stores share it): with the 8 tile stores of an odd chunk
  // issued AFTER the DMA of the next W chunk, vmcnt(8) waits for the DMA (and everything older) but not for those
  // stores - their HBM acknowledgements (~2 us under load, longer than a phase) then overlap the next phases.
  // Ragged workgroups predicate their stores (unknown count): they wait for everything.
  auto wait_dma = [&](int c_stored) {
    if (block_full && (c_stored & 1)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  for (int i = tid; i < N; i += LTHREADS) Bs[i] = to_f32<T>(bias[i]);

  // ---- output: 8 KB transpose patch per wave (64 rows x 128 B; 16-byte slot p of row r lives at slot
  //      p ^ ((r >> 1) & 7): writes 2-way, reads conflict-free) + (uniform base, 32-bit lane offset) addressing
  unsigned char* stg = &Stg[wave][0];
  unsigned char* stg_w = stg + li * 128 + 8 * hh;          // writer: row li (+32 for the second row tile)
  const unsigned stg_x = 16u * ((li >> 1) & 7);
  const int rq = lane >> 3, pq = lane & 7;                 // reader: row rq (+8 i), 16-byte piece pq
  const unsigned stg_ro = (unsigned)(rq * 128 + 16 * (pq ^ (rq >> 1)));   // rows rq + 8 i: slot also ^ 4 for odd i
  // row-major C[M][N]: row stride 2N bytes, 64-column group p at byte 128 p of the row.  planar C[N/64][M][64]:
  // row stride 128 bytes, group p is a plane of 128 M bytes - a wave's 64 x 64 tile is 8 KB CONTIGUOUS (measured:
  // contiguous runs cost ~35 us of write-back per 531 MB where row-major full lines cost ~80 us)
  const size_t ldc = planar ? 128 : (size_t)(N * 2);
  const size_t gstride = planar ? (size_t)M * 128 : 128;
  unsigned char* cblk = reinterpret_cast<unsigned char*>(C) + (size_t)blockIdx.x * LBM * ldc;
  const unsigned coff = (unsigned)((rloc + rq) * (unsigned)ldc + 16 * pq);

  f32x16 acc0, acc1;

  // ---- MFMA phase of chunk c: acc = W_chunk . A^T + bias.  The bias rides on a 25th k-step issued LAST: its W
  //      fragment is (bias[col], 0, ..) and its A fragment (1, 0, ..), both only in the hh = 0 half (k = 0), so the
  //      phase starts with a zero accumulator and no LDS round trip in front of the first MFMA.
  auto mfma_phase = [&](int c, int stage_next) {
    const unsigned char* wb = &Ws[c & 1][16 * lane];
    V8 f[3];
    f[0] = *reinterpret_cast<const V8*>(wb);
    f[1] = *reinterpret_cast<const V8*>(wb + 1024);
    const float bcol = Bs[c * LBN + li];
    if (stage_next >= 0) stage(stage_next);                // DMA issue behind the first fragment reads
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < LKS; ++s) {
      if (s + 2 < LKS) f[(s + 2) % 3] = *reinterpret_cast<const V8*>(wb + 1024 * (s + 2));
      acc0 = mfma32x32x16(f[s % 3], a0[s], acc0);          // D[col][row] += W[col][k] * A[row][k]
      acc1 = mfma32x32x16(f[s % 3], a1[s], acc1);
    }
    V8 fb, a_one;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      fb[e] = from_f32<T>((e == 0 && hh == 0) ? bcol : 0.0f);
      a_one[e] = from_f32<T>((e == 0 && hh == 0) ? 1.0f : 0.0f);
    }
    acc0 = mfma32x32x16(fb, a_one, acc0);
    acc1 = mfma32x32x16(fb, a_one, acc1);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- epilogue of chunk c: (GELU,) f16 pack, transpose patch; after every second chunk store 64 rows x 128 B ---
  auto epilogue = [&](int c) {
    const unsigned half = 64u * (c & 1);                   // which half of the 128-byte row
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x2 v[4] = {{acc0[4 * g], acc0[4 * g + 1]}, {acc0[4 * g + 2], acc0[4 * g + 3]},
                    {acc1[4 * g], acc1[4 * g + 1]}, {acc1[4 * g + 2], acc1[4 * g + 3]}};
      if (GELU) { gelu_erf2xn<LGELU_ILP>(v); if (LGELU_ILP < 4) gelu_erf2xn<LGELU_ILP>(v + 2); }
      V4 o0, o1;
      o0[0] = from_f32<T>(v[0][0]); o0[1] = from_f32<T>(v[0][1]); o0[2] = from_f32<T>(v[1][0]); o0[3] = from_f32<T>(v[1][1]);
      o1[0] = from_f32<T>(v[2][0]); o1[1] = from_f32<T>(v[2][1]); o1[2] = from_f32<T>(v[3][0]); o1[3] = from_f32<T>(v[3][1]);
      unsigned char* wp = stg_w + ((half + 16 * g) ^ stg_x);
      *reinterpret_cast<V4*>(wp) = o0;
      *reinterpret_cast<V4*>(wp + 4096) = o1;
    }
    if (!(c & 1)) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same-wave LDS write -> read (other lanes' data)
    unsigned char* cw = cblk + (size_t)(c >> 1) * gstride;
    if (block_full) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_nontemporal_store(*reinterpret_cast<const V8*>(stg + (stg_ro ^ (64u * (i & 1))) + 1024 * i),
                                    reinterpret_cast<V8*>(cw + (size_t)(8 * i) * ldc + coff));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (rloc + rq + 8 * i < mrem)
          __builtin_nontemporal_store(*reinterpret_cast<const V8*>(stg + (stg_ro ^ (64u * (i & 1))) + 1024 * i),
                                      reinterpret_cast<V8*>(cw + (size_t)(8 * i) * ldc + coff));
    }
  };

  // ---- ping-pong over the chunks.  Phase 2c: X multiplies chunk c, Y finishes chunk c-1; phase 2c+1: swapped.
  //      Chunk c+1 is DMA'd into the other buffer during phase 2c (last read in phase 2c-1) and awaited (wait_dma)
  //      before the barrier that ends phase 2c+1.
  const int nchunks = N / LBN;
  stage(0);
  wait_vm();
  __syncthreads();
  if (group_x) {
    for (int c = 0; c < nchunks; ++c) {
      mfma_phase(c, c + 1 < nchunks ? c + 1 : -1);
      phase_barrier();
      epilogue(c);
      wait_dma(c);
      phase_barrier();
    }
  } else {
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) stage(c + 1);
      if (c > 0) epilogue(c - 1);
      phase_barrier();
      mfma_phase(c, -1);
      wait_dma(c > 0 ? c - 1 : 0);
      phase_barrier();
    }
    epilogue(nchunks - 1);
  }
}

template <class T>
static void launch_linear384(const void* A, const void* W, const void* bias, void* C, int M, int N, int gelu,
                             int planar, hipStream_t s) {
  const int blocks = ceil_div(M, LBM);
  if (gelu)
    hipLaunchKernelGGL((linear_k384_kernel<T, true>), dim3(blocks), dim3(LTHREADS), 0, s, (const T*)A, (const T*)W,
                       (const T*)bias, (T*)C, M, N, planar);
  else
    hipLaunchKernelGGL((linear_k384_kernel<T, false>), dim3(blocks), dim3(LTHREADS), 0, s, (const T*)A, (const T*)W,
                       (const T*)bias, (T*)C, M, N, planar);
}

}  // namespace dss

extern "C" int dss_linear_k384(const void* A, const void* W, const void* bias, void* C, int M, int N, int gelu,
                               int out_layout, int dtype, void* stream) {
  DSS_REQUIRE(A && W && bias && C, "dss_linear_k384: null pointer");
  DSS_REQUIRE(M > 0 && N > 0 && N % (2 * dss::LBN) == 0 && N <= dss::LMAXN,
              "dss_linear_k384: need M > 0, N %% %d == 0, N <= %d (M=%d N=%d)", 2 * dss::LBN, dss::LMAXN, M, N);
  DSS_REQUIRE(out_layout == DSS_ROW_MAJOR || out_layout == DSS_PLANAR64,
              "dss_linear_k384: out_layout must be DSS_ROW_MAJOR or DSS_PLANAR64 (got %d)", out_layout);
  DSS_REQUIRE((long)dss::LBM * N * 2 < (1L << 31), "dss_linear_k384: N too large");
  hipStream_t s = (hipStream_t)stream;
  const int planar = out_layout == DSS_PLANAR64;
  switch (dtype) {
    case DSS_F16: dss::launch_linear384<dss::f16>(A, W, bias, C, M, N, gelu, planar, s); break;
    case DSS_BF16: dss::launch_linear384<dss::bf16>(A, W, bias, C, M, N, gelu, planar, s); break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_linear_k384: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  }
  DSS_CHECK_LAUNCH("linear_k384");
  return DSS_OK;
}
