// linear384.hip - Linear layers of the DINO ViTs whose reduction dimension is the embedding width:
// qkv (D -> 3D), attn.proj (D -> D), mlp.fc1 (D -> 4D, + exact GELU), for D = 384 (vits16 / vits8: the kernel was
// designed on these shapes, hence the file name) and D = 768 (vitb16 / vitb8).
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
//   * 8 waves per workgroup (two per SIMD), all on the same chunk: 50 MFMAs (fragments prefetched 2 k-steps ahead),
//     then the chunk's epilogue (VALU GELU, LDS transpose, global stores), one barrier per chunk.  The two waves of a
//     SIMD fill each other's MFMA issue gaps (one wave alone reaches ~50 % of the pipe: LDS latency behind a 2-step
//     prefetch) and drift apart enough to overlap one's epilogue with the other's MFMAs.  Rounds 1-2 ran the two
//     4-wave groups as a PING-PONG pair instead (one group multiplies while the other finishes its previous chunk, two
//     barriers per chunk): same results bit for bit, but qkv 267 -> 227 us, proj 99 -> 94, fc1+GELU 469 -> 400 without it
//     (same-process A/B, scripts/debug/linear_ab.py) - a lone MFMA wave per SIMD was the bottleneck, not the epilogue.
//   * the product is D[col][row] = W_chunk . A^T, so a lane owns token rows; the wave transposes its 64 x 64 output
//     tile (two chunks) through a private 8 KB LDS patch and stores FULL 128-byte lines, 8 rows per instruction.
//     Measured for qkv: 8-byte pieces 404 us, 64-byte half lines 310 us, full lines 268 us (same MFMA loop).
//   * D = 768 is the same kernel with ONE row tile per wave (32 rows x 768 = the same 192 VGPRs): every W fragment
//     then feeds one MFMA instead of two (twice the LDS reads per FLOP), the two accumulator chains are the even
//     and odd k-steps, and a chunk's epilogue is half as long relative to its MFMA phase.
//   * bias is the accumulators' initial value; GELU is exact-erf by Abramowitz-Stegun 7.1.28
//     (erf z = 1 - (1 + a1 z + .. + a6 z^6)^-16, |error| <= 3e-7: one v_rcp, no v_exp), written on float2 so the
//     polynomial runs on v_pk_fma_f32 / v_pk_mul_f32.
#include "common.h"
#include "kres.h"

// scripts/probes/linear_lab.hip includes this file with DSS_LIN_TIMELINE defined: wave 0 of every workgroup adds the shader
// cycles it spends in the A prologue, the MFMA phases, the epilogues and the end-of-chunk wait + barrier to dss_lin_tl.
#ifdef DSS_LIN_TIMELINE
__device__ unsigned long long dss_lin_tl[8];
#define DSS_TL_DECL unsigned long long tl_t = __builtin_readcyclecounter(), tl_acc[4] = {0, 0, 0, 0}; const unsigned long long tl_r0 = wall_clock64();
#define DSS_TL_MARK(i) { const unsigned long long n_ = __builtin_readcyclecounter(); tl_acc[i] += n_ - tl_t; tl_t = n_; }
#define DSS_TL_FLUSH if (threadIdx.x == 0) { for (int i_ = 0; i_ < 4; ++i_) atomicAdd(&dss_lin_tl[i_], tl_acc[i_]); atomicAdd(&dss_lin_tl[4], 1ull); atomicAdd(&dss_lin_tl[5], wall_clock64() - tl_r0); }
#else
#define DSS_TL_DECL
#define DSS_TL_MARK(i)
#define DSS_TL_FLUSH
#endif

namespace dss {

static constexpr int LBN = 32;            // output columns per W chunk (one MFMA column tile)
static constexpr int LGELU_ILP = 2;       // float2 pairs advanced in lockstep by the GELU (4 spills registers)

// KS = K / 16 MFMA k-steps held per token row; RT = 32-row tiles per wave.  KS * RT = 48 fragments = 192 VGPRs.
// NW = waves per workgroup.  K = 384: FOUR waves (one per SIMD) and 80 KB of LDS - TWO workgroups share a CU, the SIMD's two
// waves belong to different workgroups at different points of their row blocks (one's A prologue and epilogues can run
// under the other's MFMAs).  Measured EQUAL to the eight-wave workgroup of rounds 1-2 (qkv 258 vs 255-272 us, fc1+GELU
// 425-451 vs 435-453; scripts/debug/linear_ab.py) and kept for the smaller LDS footprint; K = 768 keeps eight waves (its
// 96 KB of W buffers admit one workgroup per CU either way).  What the lab's ablations say about this kernel (same script,
// DSS_LIN_ABL builds): no epilogue 209 us, no stores 222 us, every A fragment the SAME 16 bytes 130 us - but that last
// build also feeds the MFMAs constant operands, and on this part a dense MFMA stream runs 2.2-2.5 PFLOP/s on constant
// operands against 1.68 on random ones (profiles/r01_mfma_ceiling_probe.txt): the number is a clock effect as much as an
// A-stream effect, and neither full-line LDS-DMA loads of A nor two workgroups per CU moved the real-data time.
template <int KS, int RT, int NW> struct LinCfg {
  static_assert(KS * RT == 48 && KS % NW == 0, "the A operand of a wave is 48 fragments");
  static constexpr int WAVES = NW, THREADS = 64 * NW;
  static constexpr int K = 16 * KS;
  static constexpr int ROWS_WAVE = 32 * RT;
  static constexpr int ROWS = ROWS_WAVE * NW;               // token rows per workgroup
  static constexpr int CHUNK_BYTES = LBN * K * 2;           // 24 KB (K = 384) / 48 KB (K = 768), double buffered
  static constexpr int PATCH_BYTES = ROWS_WAVE * 128;       // transpose patch of one wave: two chunks of f16
  static constexpr int MAXN = KS == 24 ? 2048 : 3072;       // widest layer of the model family
  static constexpr int NSTORE = 4 * RT;                     // 16-byte stores per lane per finished 64-column group
};

template <class T, bool GELU, int KS, int RT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void linear_kres_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                                 const T* __restrict__ bias, T* __restrict__ C,
                                                                 int M, int N, int planar) {
  typedef typename vec8<T>::type V8;
  typedef typename vec4<T>::type V4;
  typedef LinCfg<KS, RT, NW> Cfg;
  constexpr int LK = Cfg::K, LKS = KS, LBM = Cfg::ROWS, LWAVES = NW, LTHREADS = Cfg::THREADS;
  __shared__ __attribute__((aligned(256))) unsigned char Ws[2][Cfg::CHUNK_BYTES];
  __shared__ __attribute__((aligned(256))) unsigned char Stg[LWAVES][Cfg::PATCH_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int mrem = M - blockIdx.x * LBM;                   // rows of this workgroup that exist (> 0)
  const int rloc = wave * Cfg::ROWS_WAVE;                  // this wave's first row inside the workgroup
  const bool block_full = mrem >= LBM;

  DSS_TL_DECL
  // ---- this lane's RT token rows as MFMA B-operand fragments: k = 16 s + 8 hh + e ----------------------------
  // Through the wave's own LDS patch, 64 columns at a time: LDS-DMA pieces of 8 rows x 128 B (FULL lines of A, 8 lanes per
  // row; 16-byte chunk c of row r lands at position c ^ ((r >> 1) & 7): the swizzle is applied to the source address) and
  // conflict-free ds_read_b128 of the fragments.  Rounds 1-2 loaded the fragments straight from global memory - lane
  // (li, hh) 16 bytes of row li, a wave-instruction touching 32 rows x 32 B, 1536 partial-line requests per wave where 384
  // full lines do; measured equal in time (see LinCfg), kept for the 4x fewer L2 requests.
  V8 a[RT][LKS];
  {
    typedef __attribute__((address_space(3))) void* lds3_t;
    const unsigned long long abase = (unsigned long long)(A + (long)blockIdx.x * LBM * LK);
    const unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)abase), ahi = __builtin_amdgcn_readfirstlane((unsigned)(abase >> 32));
    const unsigned char* asrc = reinterpret_cast<const unsigned char*>(((unsigned long long)ahi << 32) | alo);
    const unsigned pdst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds3_t)(&Stg[wave][0]));
    constexpr int NPIECE = Cfg::ROWS_WAVE / 8;             // 1 KB pieces per 64-column round: 8 (K = 384) / 4 (K = 768)
    constexpr int NROUND = LK / 64;
    unsigned rowoff[NPIECE];
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) {
      const int rowp = 8 * q + (lane >> 3);
      const unsigned ch = (unsigned)((lane & 7) ^ ((rowp >> 1) & 7));
#if defined(DSS_LIN_ABL) && (DSS_LIN_ABL & 1)   // lab ablation: no A stream (every piece re-reads the block's first row)
      rowoff[q] = 16u * ch;
#else
      rowoff[q] = (unsigned)min(rloc + rowp, mrem - 1) * (unsigned)(LK * 2) + 16u * ch;
#endif
    }
    const unsigned char* pw = &Stg[wave][0];
    const unsigned fsw = (unsigned)((li >> 1) & 7);
#pragma unroll
    for (int rd = 0; rd < NROUND; ++rd) {
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) {
        unsigned keep;
#if defined(DSS_LIN_ABL) && (DSS_LIN_ABL & 1)
        const unsigned off = rowoff[q];
#else
        const unsigned off = rowoff[q] + 128u * rd;
#endif
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(pdst + 1024u * q), "v"(off), "s"(asrc) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
          a[t][4 * rd + sl] = *reinterpret_cast<const V8*>(pw + (32 * t + li) * 128 + ((((unsigned)(2 * sl + hh)) ^ fsw) << 4));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments are in registers before the next round lands
    }
  }

  // ---- W chunk staging by LDS-DMA: instruction j of wave w stages k-step s = NST w + j (64 lanes x 16 B = 1 KB):
  //      lane (li, hh) fetches W[chunk col li][16 s + 8 hh .. + 8] and the hardware writes it at base + 16 * lane.
  constexpr int NST = LKS / LWAVES;                        // 3 (K = 384) / 6 (K = 768)
  const unsigned gsrc0 = (unsigned)(li * (LK * 2) + 16 * hh + 32 * (wave * NST));
  auto stage = [&](int c) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(W) + (size_t)c * Cfg::CHUNK_BYTES;  // uniform
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
  // chunk barrier WITHOUT the release fence of __syncthreads(): that fence makes the compiler drain vmcnt to 0 (the
  // tile stores!) in front of every barrier.  Nothing crosses waves through memory inside the loop except the W
  // chunks, whose arrival is awaited explicitly (wait_dma) by the issuing waves.
  auto phase_barrier = [&]() {
    asm volatile("" ::: "memory");
#ifndef DSS_LINEAR_NO_BARRIER   // lab ablation (scripts/gpu_r3.sh linear_nobar): waves drift freely, results wrong
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
  };
  // vmcnt retires in issue order (gfx9: loads, LDS-DMA and stores share it): with the 8 tile stores of an odd chunk
  // issued AFTER the DMA of the next W chunk, vmcnt(8) waits for the DMA (and everything older) but not for those
  // stores - their HBM acknowledgements (~2 us under load, longer than a phase) then overlap the next phases.
  // Ragged workgroups predicate their stores (unknown count): they wait for everything.
  auto wait_dma = [&](int c_stored) {
    if (block_full && (c_stored & 1)) {
      if (Cfg::NSTORE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  // bias of this lane's column in the chunk, fetched one chunk ahead (a plain load: hipcc waits for it at its first use,
  // the START of the next chunk's MFMA phase, right behind wait_dma + barrier where nothing younger is in flight)
  T bias_next = bias[li];
  (void)LTHREADS;

  // ---- output: transpose patch per wave (32 RT rows x 128 B; 16-byte slot p of row r lives at slot
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

  f32x16 acc0, acc1;   // RT = 2: row tiles 0 / 1.  RT = 1: even / odd k-steps of the one row tile (two MFMA chains)

  // ---- MFMA phase of chunk c: acc = W_chunk . A^T + bias.  The bias rides on a 25th k-step issued LAST: its W
  //      fragment is (bias[col], 0, ..) and its A fragment (1, 0, ..), both only in the hh = 0 half (k = 0), so the
  //      phase starts with a zero accumulator and no LDS round trip in front of the first MFMA.
  auto mfma_phase = [&](int c, int stage_next) {
    const unsigned char* wb = &Ws[c & 1][16 * lane];
    constexpr int PF = 2;                                  // W fragments in flight ahead of the MFMAs (3: measured equal)
    V8 f[PF + 1];
#pragma unroll
    for (int i = 0; i < PF; ++i) f[i] = *reinterpret_cast<const V8*>(wb + 1024 * i);
    const float bcol = to_f32<T>(bias_next);
    if ((c + 1) * LBN < N) bias_next = bias[(c + 1) * LBN + li];
    if (stage_next >= 0) stage(stage_next);                // DMA issue behind the first fragment reads
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < LKS; ++s) {
      if (s + PF < LKS) f[(s + PF) % (PF + 1)] = *reinterpret_cast<const V8*>(wb + 1024 * (s + PF));
      if (RT == 2) {
        acc0 = mfma32x32x16(f[s % (PF + 1)], a[0][s], acc0);      // D[col][row] += W[col][k] * A[row][k]
        acc1 = mfma32x32x16(f[s % (PF + 1)], a[RT - 1][s], acc1);
      } else if (s & 1) {
        acc1 = mfma32x32x16(f[s % (PF + 1)], a[0][s], acc1);
      } else {
        acc0 = mfma32x32x16(f[s % (PF + 1)], a[0][s], acc0);
      }
    }
    V8 fb, a_one;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      fb[e] = from_f32<T>((e == 0 && hh == 0) ? bcol : 0.0f);
      a_one[e] = from_f32<T>((e == 0 && hh == 0) ? 1.0f : 0.0f);
    }
    acc0 = mfma32x32x16(fb, a_one, acc0);
    if (RT == 2) acc1 = mfma32x32x16(fb, a_one, acc1);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- epilogue of chunk c: (GELU,) f16 pack, transpose patch; after every second chunk store 32 RT rows x 128 B
  auto epilogue = [&](int c) {
#if defined(DSS_LIN_ABL) && (DSS_LIN_ABL & 4)   // lab ablation: no epilogue at all (accumulators kept alive)
    asm volatile("" :: "v"(acc0), "v"(acc1));
    return;
#endif
    const unsigned half = 64u * (c & 1);                   // which half of the 128-byte row
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned char* wp = stg_w + ((half + 16 * g) ^ stg_x);
      if (RT == 2) {
        f32x2 v[4] = {{acc0[4 * g], acc0[4 * g + 1]}, {acc0[4 * g + 2], acc0[4 * g + 3]},
                      {acc1[4 * g], acc1[4 * g + 1]}, {acc1[4 * g + 2], acc1[4 * g + 3]}};
        if (GELU) { gelu_erf2xn<LGELU_ILP>(v); if (LGELU_ILP < 4) gelu_erf2xn<LGELU_ILP>(v + 2); }
        V4 o0, o1;
        o0[0] = from_f32<T>(v[0][0]); o0[1] = from_f32<T>(v[0][1]); o0[2] = from_f32<T>(v[1][0]); o0[3] = from_f32<T>(v[1][1]);
        o1[0] = from_f32<T>(v[2][0]); o1[1] = from_f32<T>(v[2][1]); o1[2] = from_f32<T>(v[3][0]); o1[3] = from_f32<T>(v[3][1]);
        *reinterpret_cast<V4*>(wp) = o0;
        *reinterpret_cast<V4*>(wp + 4096) = o1;
      } else {
        f32x2 v[2] = {{acc0[4 * g] + acc1[4 * g], acc0[4 * g + 1] + acc1[4 * g + 1]},
                      {acc0[4 * g + 2] + acc1[4 * g + 2], acc0[4 * g + 3] + acc1[4 * g + 3]}};
        if (GELU) gelu_erf2xn<2>(v);
        V4 o0;
        o0[0] = from_f32<T>(v[0][0]); o0[1] = from_f32<T>(v[0][1]); o0[2] = from_f32<T>(v[1][0]); o0[3] = from_f32<T>(v[1][1]);
        *reinterpret_cast<V4*>(wp) = o0;
      }
    }
    if (!(c & 1)) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same-wave LDS write -> read (other lanes' data)
    unsigned char* cw = cblk + (size_t)(c >> 1) * gstride;
#if defined(DSS_LIN_ABL) && (DSS_LIN_ABL & 2)   // lab ablation: no global stores
    return;
#endif
    if (block_full) {
#pragma unroll
      for (int i = 0; i < Cfg::NSTORE; ++i)
        __builtin_nontemporal_store(*reinterpret_cast<const V8*>(stg + (stg_ro ^ (64u * (i & 1))) + 1024 * i),
                                    reinterpret_cast<V8*>(cw + (size_t)(8 * i) * ldc + coff));
    } else {
#pragma unroll
      for (int i = 0; i < Cfg::NSTORE; ++i)
        if (rloc + rq + 8 * i < mrem)
          __builtin_nontemporal_store(*reinterpret_cast<const V8*>(stg + (stg_ro ^ (64u * (i & 1))) + 1024 * i),
                                      reinterpret_cast<V8*>(cw + (size_t)(8 * i) * ldc + coff));
    }
  };

  // ---- the chunks.  Chunk c + 1 is DMA'd into the other buffer at the start of chunk c's MFMA phase (that buffer was
  //      last read in chunk c - 1, behind the barrier that ended it) and awaited (wait_dma) before the barrier that ends
  //      chunk c.
  const int nchunks = N / LBN;
  stage(0);
  wait_vm();
  __syncthreads();
  DSS_TL_MARK(0)
  for (int c = 0; c < nchunks; ++c) {
    mfma_phase(c, c + 1 < nchunks ? c + 1 : -1);
    DSS_TL_MARK(1)
    epilogue(c);
    DSS_TL_MARK(2)
    wait_dma(c);
    phase_barrier();
    DSS_TL_MARK(3)
  }
  DSS_TL_FLUSH
}

template <class T, int KS, int RT, int NW>
static void launch_linear_kres(const void* A, const void* W, const void* bias, void* C, int M, int N, int gelu,
                               int planar, hipStream_t s) {
  const int blocks = ceil_div(M, LinCfg<KS, RT, NW>::ROWS);
  if (gelu)
    hipLaunchKernelGGL((linear_kres_kernel<T, true, KS, RT, NW>), dim3(blocks), dim3(64 * NW), 0, s, (const T*)A,
                       (const T*)W, (const T*)bias, (T*)C, M, N, planar);
  else
    hipLaunchKernelGGL((linear_kres_kernel<T, false, KS, RT, NW>), dim3(blocks), dim3(64 * NW), 0, s, (const T*)A,
                       (const T*)W, (const T*)bias, (T*)C, M, N, planar);
}

template <int KS, int RT, int NW>
static int linear_kres(const char* name, const void* A, const void* W, const void* bias, void* C, int M, int N,
                       int gelu, int out_layout, int dtype, void* stream) {
  typedef LinCfg<KS, RT, NW> Cfg;
  DSS_REQUIRE(A && W && bias && C, "%s: null pointer", name);
  DSS_REQUIRE(M > 0 && N > 0 && N % (2 * LBN) == 0 && N <= Cfg::MAXN, "%s: need M > 0, N %% %d == 0, N <= %d (M=%d N=%d)",
              name, 2 * LBN, Cfg::MAXN, M, N);
  DSS_REQUIRE(out_layout == DSS_ROW_MAJOR || out_layout == DSS_PLANAR64,
              "%s: out_layout must be DSS_ROW_MAJOR or DSS_PLANAR64 (got %d)", name, out_layout);
  hipStream_t s = (hipStream_t)stream;
  const int planar = out_layout == DSS_PLANAR64;
  switch (dtype) {
    case DSS_F16: launch_linear_kres<f16, KS, RT, NW>(A, W, bias, C, M, N, gelu, planar, s); break;
    case DSS_BF16: launch_linear_kres<bf16, KS, RT, NW>(A, W, bias, C, M, N, gelu, planar, s); break;
    default: return fail(DSS_ERR_BAD_ARG, "%s: dtype must be DSS_F16 or DSS_BF16 (got %d)", name, dtype);
  }
  DSS_CHECK_LAUNCH(name);
  return DSS_OK;
}

}  // namespace dss

extern "C" int dss_linear_k384(const void* A, const void* W, const void* bias, void* C, int M, int N, int gelu,
                               int out_layout, int dtype, void* stream) {
  return dss::linear_kres<24, 2, 4>("dss_linear_k384", A, W, bias, C, M, N, gelu, out_layout, dtype, stream);
}

extern "C" int dss_linear_k768(const void* A, const void* W, const void* bias, void* C, int M, int N, int gelu,
                               int out_layout, int dtype, void* stream) {
  return dss::linear_kres<48, 1, 8>("dss_linear_k768", A, W, bias, C, M, N, gelu, out_layout, dtype, stream);
}
