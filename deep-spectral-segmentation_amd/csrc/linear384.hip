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
//   * round 4: the LayerNorm in front of qkv / fc1 (and the residual add in front of IT) is the kernel's A prologue
//     (dss_lnlinear_k384 / _k768, LNM != 0 below): the standalone pass read x f32 + the pending branch output, wrote x f32
//     + h f16, and this kernel read h again - 23 launches of 1.2 GB per forward that only re-read what a wave is about to
//     hold.  Here a wave walks its RT x K/32 UNITS of 32 rows x 32 columns: x arrives by LDS-DMA as full 128-byte lines
//     (4 KB per unit, a ring of slots in the - still idle - W double buffer), the residual tile (32 rows x 64 columns of
//     f16: two units) in the wave's transpose patch; the lane reads its fragment-shaped share, adds, accumulates
//     pivot-shifted first and second moments of ITS two rows (a lane owns whole half rows: no butterfly, one lane^32
//     exchange at the end), packs the sum MINUS THE ROW'S PIVOT (round 5: a robust typical value of the row, so that the operand's
//     rounding error scales with the row's spread and not with |x| - rows whose mean is far from zero) into the resident A
//     fragments, writes the f32 sums back into the slot and stores them as full lines.  The fragments stay un-normalised
//     (f16(x - pivot): one rounding): (x - mean) rstd never exists - the (mean - pivot) and
//     sigma corrections are one fp32 MFMA k-step per chunk against the caller's table aux[col] = (-sum_k Wg, b') and the
//     epilogue multiplies by rstd (gamma and beta are folded into W and the table by dss_lnlinear_prepare).
//   * round 4: the hooked block's K projection is the same body in a hand-over mode (kfeat_kres_kernel,
//     dss_lnlinear_kfeatures_k384; see KfOut): read-only prologue, fp32 features from the accumulators + f16 copy + inverse
//     row norms, CLS rows dropped.
//   * K = 384 runs FOUR waves per workgroup and two workgroups per CU (see LinCfg); K = 768 eight waves, one workgroup.
//   * this file builds exactly ONE schedule.  The lab variants of rounds 2-4 (ablations, plain prefetch, one tile per wave with
//     three waves per SIMD, one wave per SIMD with the epilogue inside the next chunk's MFMAs, staggered workgroups, scalar GELU;
//     results in profiles/r0N_linear_lab.txt, none shipped) live in scripts/probes/linear384_r4_lab.hip.  The only hook left is
//     DSS_LIN_TIMELINE (instrumentation of THIS schedule, scripts/probes/linear_lab.hip); it compiles to nothing in the library.
#include "common.h"
#include "kres.h"
#include <type_traits>
#include <utility>

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

// LDS fragment read / counted wait with the order fixed by the source (see mfma_phase)
template <int OFF, class V> __device__ __forceinline__ void lds_read_b128_at(V& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N, class V> __device__ __forceinline__ void lds_wait_for(V& v) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// Issue order of the vector-memory instructions of the LayerNorm prologue (every event below is 4 instructions of one
// wave; gfx9 retires loads, LDS-DMA and stores through ONE in-order vmcnt):  Pt(0), X(0..NXS-1), Pt(1);  then after unit
// v has been processed: ST(v) [the x write-back], X(v + NXS), and behind an odd v the residual tile Pt(v/2 + 2).
// after(u) = instructions issued behind the younger of X(u) / Pt(u/2) when unit u is waited for = its s_waitcnt vmcnt.
template <int NUNIT, int NXS, bool RES, bool ST = RES> struct LnSched {   // ST: the x write-back exists (not in the hand-over mode)
  static constexpr int after(int u) {
    int n = 0, px[64] = {}, pp[40] = {};
    if (RES) { pp[0] = n; n += 4; }
    for (int i = 0; i < NXS && i < NUNIT; ++i) { px[i] = n; n += 4; }
    if (RES && NUNIT > 2) { pp[1] = n; n += 4; }
    for (int v = 0; v < NUNIT; ++v) {
      if (v == u) {
        int last = px[u];
        if (RES && pp[u >> 1] > last) last = pp[u >> 1];
        return n - (last + 4);
      }
      if (ST) n += 4;
      if (v + NXS < NUNIT) { px[v + NXS] = n; n += 4; }
      if (RES && (v & 1) && (v >> 1) + 2 < NUNIT / 2) { pp[(v >> 1) + 2] = n; n += 4; }
    }
    return 0;
  }
};

// KS = K / 16 MFMA k-steps held per token row; RT = 32-row tiles per wave.  KS * RT = 48 fragments = 192 VGPRs.
// NW = waves per workgroup.  K = 384: FOUR waves (one per SIMD) and 80 KB of LDS - TWO workgroups share a CU, the SIMD's two
// waves belong to different workgroups at different points of their row blocks (one's A prologue and epilogues can run
// under the other's MFMAs).  Measured EQUAL to the eight-wave workgroup of rounds 1-2 (qkv 258 vs 255-272 us, fc1+GELU
// 425-451 vs 435-453; scripts/debug/linear_ab.py) and kept for the smaller LDS footprint; K = 768 keeps eight waves (its
// 96 KB of W buffers admit one workgroup per CU either way).  What the lab's ablations say about this kernel (same script,
// ablation builds of the lab snapshot): no epilogue 209 us, no stores 222 us, every A fragment the SAME 16 bytes 130 us - but that last
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

// LNM: 0 = A [M, K] is given;  1 = A = LN(x) without affine;  2 = x += res in place first, then A = LN(x)  (x f32 [M, K];
// res [M, K] of T, element (r, c) at r * r_ld + (c / 64) * r_plane + c % 64: row-major (K, 64) or DSS_PLANAR64 (64, 64 M)).
// MODE 0: the Linear layer (output C, row-major or DSS_PLANAR64).
// MODE 2 (K-feature hand-over, dss_lnlinear_kfeatures): the output of the LAST block's K projection leaves as what the
// caller and the affinity build need - token rows b * Tn + t, t >= 1, go to row b * (Tn - 1) + t - 1 of k32 (fp32, straight
// from the accumulators: 16-byte pieces, a 32-column chunk of a row is one 128-byte line written by one wave), of C = k16
// (through the transpose patch, as every other output) and rnorm = 1 / max(|k16 row|, eps); CLS rows are computed and dropped.
// MODE 4 (patch embedding, dss_patch_embed_p16): DINO's PatchEmbed Conv2d(3, D, 16, 16) + `x = tokens + pos_embed[1:]` straight from
// the u8 image: a lane's operand row is one 16 x 16 x 3 patch, gathered as 8-byte pieces of its 48-byte pixel rows (k order (py, px,
// c); ToTensor / Normalize are folded into the weight and bias by the caller), the operand is pixel - 128 (-128..127: exact in f16 /
// bf16, and centred so that the rounding of the folded weight multiplies a deviation, not the 0..255 level); the
// epilogue adds the position embedding of the patch and writes fp32 rows b (Np + 1) + n + 1 of the residual stream - no patchify
// pass, no f16 token tensor, no position-embedding pass.  k32 = x, Tn = Np, img / pos / H / W / Wp as named.
struct KfOut { float* k32; float* rnorm; int Tn; float eps; const unsigned char* img; const float* pos; int H, W, Wp; };

template <class T, int GELU, int KS, int RT, int NW, int LNM, int MODE>
__device__ __forceinline__ void linear_kres_body(const T* __restrict__ A, float* __restrict__ X,
                                                 const T* __restrict__ R, long r_ld, long r_plane, float eps,
                                                 const T* __restrict__ W,
                                                 const T* __restrict__ bias, const float* __restrict__ aux,
                                                 T* __restrict__ C, int M, int N, int planar, const KfOut kf) {
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
  float am[RT];                                            // LNM != 0: A side of the correction k-step (mean | sigma), per row tile
  if constexpr (MODE == 4) {
    static_assert(RT == 1 && LK == 768, "one 16 x 16 x 3 patch per operand row");
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const unsigned gp = (unsigned)min((int)blockIdx.x * LBM + rloc + li, M - 1);   // this lane's patch (M = B Np of them)
    const unsigned pb = gp / (unsigned)kf.Tn, pn = gp - pb * (unsigned)kf.Tn;
    const unsigned py0 = (pn / (unsigned)kf.Wp) * 16u, px0 = (pn - (pn / (unsigned)kf.Wp) * (unsigned)kf.Wp) * 16u;
    const unsigned w3 = (unsigned)kf.W * 3u;
    const unsigned char* psrc = kf.img + ((size_t)(pb * (unsigned)kf.H + py0) * kf.W + px0) * 3;
    static_for<LKS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;                   // k = 16 s + 8 hh .. + 8: pixel row k / 48 of the patch, byte k % 48 of it
      constexpr int r0 = (16 * s) / 48, o0 = (16 * s) % 48, r1 = (16 * s + 8) / 48, o1 = (16 * s + 8) % 48;
      const unsigned off = hh ? (unsigned)r1 * w3 + (unsigned)o1 : (unsigned)r0 * w3 + (unsigned)o0;
      const u32x2v raw = *reinterpret_cast<const u32x2v*>(psrc + off);
      V8 fr;
#pragma unroll
      for (int e = 0; e < 8; ++e) fr[e] = from_f32<T>((float)((raw[e >> 2] >> (8 * (e & 3))) & 0xffu) - 128.0f);   // centred: see KfOut
      a[0][s] = fr;
    });
  } else if constexpr (LNM == 0) {
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
      rowoff[q] = (unsigned)min(rloc + rowp, mrem - 1) * (unsigned)(LK * 2) + 16u * ch;
    }
    // Two landing buffers - the wave's patch and its share of the (still idle) W double buffer - so that round rd + 1 is in
    // flight while round rd's fragments are read: a round used to cost a full DMA round trip (issue, vmcnt(0), read), six
    // (twelve) of them back to back were 25 % of the N = 384 kernel.
    constexpr int WS_SHARE = 2 * Cfg::CHUNK_BYTES / LWAVES;
    static_assert(WS_SHARE >= NPIECE * 1024, "the wave's share of the W buffers holds one 64-column round");
    const unsigned char* pbuf[2] = {&Stg[wave][0], &Ws[0][0] + wave * WS_SHARE};
    const unsigned pdst2[2] = {pdst, (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds3_t)(&Ws[0][0] + wave * WS_SHARE))};
    const unsigned fsw = (unsigned)((li >> 1) & 7);
    auto issue_round = [&](int rd) {
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) {
        unsigned keep;
        const unsigned off = rowoff[q] + 128u * rd;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(pdst2[rd & 1] + 1024u * q), "v"(off), "s"(asrc) : "memory");
      }
    };
    issue_round(0);
    if (NROUND > 1) issue_round(1);
    static_for<NROUND>([&](auto rc) {
      constexpr int rd = decltype(rc)::value;
      if constexpr (rd + 1 < NROUND) wait_vmcnt<NPIECE>();    // round rd has landed, rd + 1 may still be in flight
      else wait_vmcnt<0>();
      const unsigned char* pw = pbuf[rd & 1];
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
          a[t][4 * rd + sl] = *reinterpret_cast<const V8*>(pw + (32 * t + li) * 128 + ((((unsigned)(2 * sl + hh)) ^ fsw) << 4));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments are in registers before the buffer is reused
      if constexpr (rd + 2 < NROUND) issue_round(rd + 2);
    });
    __syncthreads();                                           // the W double buffer returns to its owner
  } else {
    // ---- LayerNorm prologue: x (+= res) -> statistics -> normalised A fragments (see the file header) ---------------------
    typedef __attribute__((address_space(3))) void* lds3_t;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    constexpr bool RES = LNM == 2;
    constexpr int NCB = LK / 32, NUNIT = RT * NCB;          // 32-column blocks per row, units per wave
    constexpr int WS_SHARE = 2 * Cfg::CHUNK_BYTES / LWAVES;  // this wave's share of the idle W double buffer: 12 KB
    constexpr int NSLOT = (WS_SHARE + Cfg::PATCH_BYTES) / 4096;
    constexpr int NXS = RES ? NSLOT - 2 : NSLOT;             // x slots (K = 384: 3 / 5; K = 768: 2 / 4); 2 residual slots
    static_assert(WS_SHARE % 4096 == 0 && Cfg::PATCH_BYTES % 4096 == 0 && NXS >= 2 && NCB % 2 == 0 && NUNIT <= 48, "LN prologue layout");
    constexpr bool XST = RES && MODE != 2;                  // the hand-over kernel is the stream's LAST reader: x + res is used, not stored
    typedef LnSched<NUNIT, NXS, RES, XST> Sched;
    unsigned char* const ws_share = &Ws[0][0] + wave * WS_SHARE;
    unsigned char* const patch = &Stg[wave][0];
    auto slot_ptr = [&](int i) -> unsigned char* { return i * 4096 < WS_SHARE ? ws_share + i * 4096 : patch + (i * 4096 - WS_SHARE); };
    const size_t row0 = (size_t)blockIdx.x * LBM;
    const unsigned char* const xblk = reinterpret_cast<const unsigned char*>(X) + row0 * (size_t)(LK * 4);
    const unsigned char* const rblk = reinterpret_cast<const unsigned char*>(R) + row0 * (size_t)r_ld * 2;
    const unsigned r_ldb = (unsigned)r_ld * 2u;
    const unsigned fsw16 = 16u * (unsigned)((li >> 1) & 7);
    // one DMA piece = 8 rows x 128 B: lane l -> row 8 q + (l >> 3), 16-byte chunk (l & 7) ^ swizzle(row) (source-side swizzle)
    const int prow = lane >> 3;
    const unsigned pch16[2] = {16u * (unsigned)((lane & 7) ^ (lane >> 4)), 16u * (unsigned)((lane & 7) ^ (4 + (lane >> 4)))};
    auto dma = [&](unsigned lds_addr, unsigned voff, const unsigned char* sbase) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
    };
    auto uniform_ptr = [&](const unsigned char* p) -> const unsigned char* {
      const unsigned long long v = (unsigned long long)p;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
    };
    auto rowc = [&](int t, int q) { return (unsigned)min(rloc + 32 * t + 8 * q + prow, mrem - 1); };
    auto issue_x = [&](int u) {                               // unit u = (t, cb): 32 rows x 32 columns of f32
      const int t = u / NCB, cb = u % NCB;
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds3_t)slot_ptr(u % NXS));
      const unsigned char* src = uniform_ptr(xblk + 128 * cb);
#pragma unroll
      for (int q = 0; q < 4; ++q) dma(dst + 1024u * q, rowc(t, q) * (unsigned)(LK * 4) + pch16[q & 1], src);
    };
    auto issue_p = [&](int j) {                               // residual tile j = (t, 64-column group): 32 rows x 128 B
      const int t = (2 * j) / NCB, g = ((2 * j) % NCB) >> 1;
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds3_t)slot_ptr(NXS + (j & 1)));
      const unsigned char* src = uniform_ptr(rblk + (size_t)g * (size_t)r_plane * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) dma(dst + 1024u * q, rowc(t, q) * r_ldb + pch16[q & 1], src);
    };
    float piv[RT], s1[RT], s2[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { piv[t] = 0.f; s1[t] = 0.f; s2[t] = 0.f; }
    if (RES) issue_p(0);
#pragma unroll
    for (int i = 0; i < NXS && i < NUNIT; ++i) issue_x(i);
    if (RES && NUNIT > 2) issue_p(1);
    static_for<NUNIT>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      constexpr int t = u / NCB, cb = u % NCB, o = cb & 1;
      if (block_full) wait_vmcnt<Sched::after(u)>();
      else wait_vmcnt<0>();                                   // a ragged block predicates its stores: unknown counts
      unsigned char* xs = slot_ptr(u % NXS);
      unsigned char* xr = xs + li * 128;
      f32x4v v0[2], v1[2];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        v0[sl] = *reinterpret_cast<const f32x4v*>(xr + ((16u * (4 * sl + 2 * hh)) ^ fsw16));
        v1[sl] = *reinterpret_cast<const f32x4v*>(xr + ((16u * (4 * sl + 2 * hh + 1)) ^ fsw16));
      }
      if constexpr (RES) {
        const unsigned char* pr = slot_ptr(NXS + ((u >> 1) & 1)) + li * 128;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const V8 pv = *reinterpret_cast<const V8*>(pr + ((16u * (4 * o + 2 * sl + hh)) ^ fsw16));
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[sl][e] += to_f32<T>(pv[e]); v1[sl][e] += to_f32<T>(pv[4 + e]); }
        }
      }
      if constexpr (cb == 0) {
        // the ROW's pivot (round 5; both lanes of a row use the hh = 0 lane's): the median of the medians of three column triples
        // (columns 0-2, 4-6, 16-18: Tukey's ninther) - a typical value of the row unless FOUR of those nine columns are outlier
        // channels (a pivot taken from a massive-activation channel would put every other column 2^-11 |outlier| from its value;
        // trained ViTs have two or three such channels), within ~0.4 sigma of the row's centre otherwise
        const float g0 = __builtin_amdgcn_fmed3f(v0[0][0], v0[0][1], v0[0][2]), g1 = __builtin_amdgcn_fmed3f(v1[0][0], v1[0][1], v1[0][2]);
        const float g2 = __builtin_amdgcn_fmed3f(v0[1][0], v0[1][1], v0[1][2]);
        const unsigned pm = __float_as_uint(__builtin_amdgcn_fmed3f(g0, g1, g2));
        const auto pr = __builtin_amdgcn_permlane32_swap(pm, pm, false, false);
        const unsigned plo = pr[0];                              // (element 0 = the lower half's value in every lane; copied to a
        piv[t] = __uint_as_float(plo);                           //  scalar first: see half_pair_max in attention.hip)
      }
      float u1[2] = {0.f, 0.f}, u2[2] = {0.f, 0.f};            // this unit's moments: two short chains per k-step, not one of 32
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        V8 fr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = v0[sl][e] - piv[t], d1 = v1[sl][e] - piv[t];
          u1[sl] += d0 + d1;
          u2[sl] = fmaf(d0, d0, fmaf(d1, d1, u2[sl]));
          fr[e] = from_f32<T>(d0);                               // the operand is x - pivot: its rounding error scales with the
          fr[4 + e] = from_f32<T>(d1);                           // row's spread, not with |x| (rows whose mean is far from zero)
        }
        asm volatile("" : "+v"(fr));                           // packed HERE (hipcc otherwise carries the f32 values to the end)
        a[t][2 * cb + sl] = fr;
      }
      s1[t] += u1[0] + u1[1];
      s2[t] += u2[0] + u2[1];
      asm volatile("" : "+v"(s1[t]), "+v"(s2[t]));             // ... and the moments are final HERE, not 20 units later
      if constexpr (XST) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {                       // the sums go back into the slot (same lanes, same places) ...
          *reinterpret_cast<f32x4v*>(xr + ((16u * (4 * sl + 2 * hh)) ^ fsw16)) = v0[sl];
          *reinterpret_cast<f32x4v*>(xr + ((16u * (4 * sl + 2 * hh + 1)) ^ fsw16)) = v1[sl];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ... and leave as full lines: piece q = rows 8 q .. 8 q + 7
        unsigned char* xw = const_cast<unsigned char*>(xblk) + 128 * cb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4v lin = *reinterpret_cast<const f32x4v*>(xs + 1024 * q + 16 * lane);
          if (block_full || rloc + 32 * t + 8 * q + prow < mrem)
            *reinterpret_cast<f32x4v*>(xw + rowc(t, q) * (unsigned)(LK * 4) + pch16[q & 1]) = lin;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot is free: every LDS read of it has returned
      if (u + NXS < NUNIT) issue_x(u + NXS);
      if (RES && (u & 1) && (u >> 1) + 2 < NUNIT / 2) issue_p((u >> 1) + 2);
    });
    // statistics: this lane holds K/2 elements of each of its RT rows (pivot-shifted moments), lane ^ 32 the other half.
    // The fragments stay UN-normalised (f16(x - pivot): one rounding).  (x - mean) rstd never exists: with sw[col] = sum_k W[col][k]
    // and sigma = 1 / rstd, out = rstd (acc - (mean - pivot) sw[col] + sigma b[col]) - the bracket's two corrections are ONE fp32 MFMA
    // (v_mfma_f32_32x32x2_f32: k = 0 multiplies -sw[col] with (mean - pivot)[row], k = 1 b[col] with sigma[row]; exact fp32
    // products) where the plain kernel has its bias k-step, the table aux[col] = (-sw, b) is the caller's
    // (dss_lnlinear_prepare); the epilogue multiplies by rstd.
    constexpr float HN = (float)(LK / 2);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const float mh = piv[t] + s1[t] * (1.0f / HN), m2h = fmaxf(s2[t] - s1[t] * s1[t] * (1.0f / HN), 0.f);
      const float mo = __shfl_xor(mh, 32, 64), m2o = __shfl_xor(m2h, 32, 64);
      const float mean = 0.5f * (mh + mo), dl = mo - mh;
      const float var = (m2h + m2o + dl * dl * (0.5f * HN)) * (1.0f / (float)LK);
      am[t] = hh ? (var + eps) * rsqrtf(var + eps) : mean - piv[t];   // A side of the correction k-step: k = 0 mean - pivot, k = 1 sigma
    }
    __syncthreads();                                           // the W double buffer returns to its owner
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
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // vmcnt retires in issue order (gfx9: loads, LDS-DMA and stores share it): with the 8 tile stores of an odd chunk
  // issued AFTER the DMA of the next W chunk, vmcnt(8) waits for the DMA (and everything older) but not for those
  // stores - their HBM acknowledgements (~2 us under load, longer than a phase) then overlap the next phases.
  // Ragged workgroups predicate their stores (unknown count): they wait for everything.
  auto wait_dma = [&](int c_stored) {
    if (MODE != 2 && MODE != 4 && block_full && (c_stored & 1)) {          // (the hand-over mode predicates its stores per row: unknown counts)
      if (Cfg::NSTORE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  // bias of this lane's column in the chunk, fetched one chunk ahead (a plain load: hipcc waits for it at its first use,
  // the START of the next chunk's MFMA phase, right behind wait_dma + barrier where nothing younger is in flight)
  T bias_next = from_f32<T>(0.f);
  if constexpr (LNM == 0) bias_next = bias[li];
  // LNM: aux[col] = (-sw, b) of this lane's column (hh = 0 / 1), the same way - a plain load one chunk ahead.  (Round 4 fetched it
  // at the start of the chunk's own MFMA phase from inline asm and waited for it by count in front of the last MFMA: hipcc is
  // free to COPY the destination register between the two asm statements, and in the last chunk's branch it did, in front of
  // the wait - a stale correction term in the last 32 columns of a few workgroups whenever the load took longer than the
  // MFMA phase, 1-8 launches in 300; scripts/debug/lnlinear_stress.py.  A load the compiler issues is one it waits for.)
  auto aux_lane = [&](int c) -> float {
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));   // fresh lane id: no offset register is held
    const unsigned aoff = ((l & 31u) << 3) | ((l >> 5) << 2);
    return *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(aux + 2 * c * LBN) + aoff);
  };
  float wc_next = 0.f;
  if constexpr (LNM != 0) wc_next = aux_lane(0);
  (void)LTHREADS;

  // ---- output: transpose patch per wave (32 RT rows x 128 B; 16-byte slot p of row r lives at slot
  //      p ^ ((r >> 1) & 7): writes 2-way, reads conflict-free) + (uniform base, 32-bit lane offset) addressing
  // (the lane-derived addresses of the epilogue are rebuilt in it from a fresh lane id: six registers that would
  //  otherwise be held through the MFMA phases, which need them for the third W-fragment buffer)
  unsigned char* stg = &Stg[wave][0];
  // row-major C[M][N]: row stride 2N bytes, 64-column group p at byte 128 p of the row.  planar C[N/64][M][64]:
  // row stride 128 bytes, group p is a plane of 128 M bytes - a wave's 64 x 64 tile is 8 KB CONTIGUOUS (measured:
  // contiguous runs cost ~35 us of write-back per 531 MB where row-major full lines cost ~80 us)
  const size_t ldc = planar ? 128 : (size_t)(N * 2);
  const size_t gstride = planar ? (size_t)M * 128 : 128;
  unsigned char* cblk = reinterpret_cast<unsigned char*>(C) + (size_t)blockIdx.x * LBM * ldc;

  f32x16 acc0, acc1;   // RT = 2: row tiles 0 / 1.  RT = 1: even / odd k-steps of the one row tile (two MFMA chains)

  // ---- MFMA phase of chunk c: acc = W_chunk . A^T + bias.  The bias rides on a 25th k-step issued LAST: its W
  //      fragment is (bias[col], 0, ..) and its A fragment (1, 0, ..), both only in the hh = 0 half (k = 0), so the
  //      phase starts with a zero accumulator and no LDS round trip in front of the first MFMA.
  auto mfma_phase = [&](int c, int stage_next) {
    // W fragments come out of LDS TWO k-steps ahead of the MFMAs that use them, and the order is pinned in assembly.  As plain
    // C++ loads hipcc regroups them - two reads, then lgkmcnt waits straight behind them, then four MFMAs - so that every
    // fourth MFMA waits out an LDS round trip: a wave alone on its SIMD (its partner in its epilogue) then runs at ~45 % of
    // the matrix pipe and the partner's epilogue buys nothing (round-3 timeline: an MFMA phase takes 3 340 cycles whether
    // or not the other wave is in one).  LDS returns in order: the wait for the fragment of step s leaves the reads of
    // steps s + 1, s + 2 outstanding (counted waits; the fragment passes through the wait as an operand, which is what keeps
    // its MFMAs behind it).
    const unsigned wbase = (unsigned)(size_t)(lds_ptr_t)(&Ws[c & 1][0]) + 16u * (unsigned)lane;
    constexpr int PF = 2;
    V8 f[PF + 1];
    static_for<PF>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_read_b128_at<1024 * i>(f[i], wbase); });
    const float bcol = to_f32<T>(bias_next);
    float wcorr = wc_next;
    if constexpr (LNM == 0) {
      if ((c + 1) * LBN < N) bias_next = bias[(c + 1) * LBN + li];
    } else {
      // this chunk's correction term arrived a chunk ago; the use pinned HERE is where hipcc puts its wait for that load - behind
      // wait_dma + the barrier, nothing younger in flight - and not in front of the last MFMA, where the next W chunk's DMA pieces are
      asm volatile("" : "+v"(wcorr));
      if ((c + 1) * LBN < N) wc_next = aux_lane(c + 1);
    }
    if (stage_next >= 0) stage(stage_next);                // DMA issue behind the first fragment reads
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    __builtin_amdgcn_s_setprio(1);
    static_for<LKS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s + PF < LKS) lds_read_b128_at<1024 * (s + PF)>(f[(s + PF) % (PF + 1)], wbase);
      lds_wait_for<(s + PF < LKS ? PF : LKS - 1 - s)>(f[s % (PF + 1)]);
      if (RT == 2) {
        acc0 = mfma32x32x16(f[s % (PF + 1)], a[0][s], acc0);      // D[col][row] += W[col][k] * A[row][k]
        acc1 = mfma32x32x16(f[s % (PF + 1)], a[RT - 1][s], acc1);
      } else if (s & 1) {
        acc1 = mfma32x32x16(f[s % (PF + 1)], a[0][s], acc1);
      } else {
        acc0 = mfma32x32x16(f[s % (PF + 1)], a[0][s], acc0);
      }
    });
    if constexpr (LNM == 0) {
      V8 fb, a_one;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        fb[e] = from_f32<T>((e == 0 && hh == 0) ? bcol : 0.0f);
        a_one[e] = from_f32<T>((e == 0 && hh == 0) ? 1.0f : 0.0f);
      }
      acc0 = mfma32x32x16(fb, a_one, acc0);
      if (RT == 2) acc1 = mfma32x32x16(fb, a_one, acc1);
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wcorr, am[0], acc0, 0, 0, 0);
      if (RT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wcorr, am[RT - 1], acc1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- K-feature hand-over (MODE == 2, see KfOut): this lane's two accumulator rows -> output rows, their sums of squares
  float kss[RT];
  unsigned kmagic = 0;                                     // gr / Tn = umulhi(gr, kmagic) for gr * Tn < 2^32 (host-checked)
  if constexpr (MODE == 4) kmagic = __builtin_amdgcn_readfirstlane(0xFFFFFFFFu / (unsigned)kf.Tn + 1u);
  if constexpr (MODE == 2) {
    kmagic = __builtin_amdgcn_readfirstlane(0xFFFFFFFFu / (unsigned)kf.Tn + 1u);
#pragma unroll
    for (int t = 0; t < RT; ++t) kss[t] = 0.f;
  }
  // output row of accumulator row (tile t, lane row l31), rebuilt where it is used (a register pair held through the MFMA
  // phases otherwise): -1 = CLS or past M
  auto kf_row = [&](int t, unsigned l31) -> int {           // (one division: rows l31 and l31 + 32 straddle at most one image boundary)
    const unsigned gr0 = (unsigned)blockIdx.x * LBM + (unsigned)__builtin_amdgcn_readfirstlane(rloc) + l31, gr = gr0 + 32 * t;
    const unsigned b0 = __umulhi(gr0, kmagic);
    const unsigned cls0 = b0 * (unsigned)kf.Tn, cls1 = cls0 + (unsigned)kf.Tn;
    return ((int)gr < M && gr != cls0 && gr != cls1) ? (int)(gr - b0 - 1u - (gr >= cls1 ? 1u : 0u)) : -1;
  };
  auto kf_finish = [&]() {
    unsigned fl;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(fl));   // fresh lane id (see the epilogue)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const unsigned u = __float_as_uint(kss[t]);
      const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // the other half of the row's columns: lane ^ 32
      const unsigned lo = r[0], hi = r[1];
      const float tot = __uint_as_float(lo) + __uint_as_float(hi);
      const int orow = kf_row(t, fl & 31u);
      if (fl < 32u && orow >= 0) kf.rnorm[orow] = 1.0f / fmaxf(sqrtf(tot), kf.eps);
    }
  };

  // ---- epilogue of chunk c: (GELU,) f16 pack, transpose patch; after every second chunk store 32 RT rows x 128 B
  auto epilogue = [&](int c) {
    unsigned el;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));   // opaque: nothing below is hoisted
    const unsigned eli = el & 31u, ehh = el >> 5;
    unsigned char* stg_w = stg + eli * 128 + 8 * ehh;      // writer: row li (+32 for the second row tile)
    const unsigned stg_x = 16u * ((eli >> 1) & 7);
    const int rq = (int)(el >> 3), pq = (int)(el & 7);     // reader: row rq (+8 i), 16-byte piece pq
    const unsigned stg_ro = (unsigned)(rq * 128 + 16 * (pq ^ (rq >> 1)));   // rows rq + 8 i: slot also ^ 4 for odd i
    const unsigned coff = (unsigned)((rloc + rq) * (unsigned)ldc + 16 * pq);
    if constexpr (LNM != 0) {                              // acc *= 1 / sigma, in place: sigma lives in the hh = 1 lane of the pair
#pragma unroll                                             // (rstd kept per row tile would be registers the GELU does not have)
      for (int t = 0; t < RT; ++t) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(am[t]), __float_as_uint(am[t]), false, false);
        const unsigned sg = r[1];
        const float rs = __builtin_amdgcn_rcpf(__uint_as_float(sg));
        if (RT == 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) { acc0[i] *= rs; acc1[i] *= rs; }
        } else if (t == 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc0[i] *= rs;
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc1[i] *= rs;
        }
      }
    }
    if constexpr (MODE == 4) {
      // tokens + position embedding -> fp32 rows of the residual stream, straight from the accumulators (no patch, no f16 output)
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      const unsigned gp = blockIdx.x * LBM + (unsigned)__builtin_amdgcn_readfirstlane(rloc) + eli;
      const unsigned pb = __umulhi(gp, kmagic);
      const unsigned coloff = (unsigned)(c * LBN * 4) + 16u * ehh;
      const unsigned char* posr = reinterpret_cast<const unsigned char*>(kf.pos) + (gp - pb * (unsigned)kf.Tn) * (unsigned)(N * 4) + coloff;
      unsigned char* xr = reinterpret_cast<unsigned char*>(kf.k32) + (gp + pb + 1u) * (unsigned)(N * 4) + coloff;
      // the two k-step chains are joined FIRST (and pinned there): the 16 registers of the odd chain are then free for the
      // position rows - with the loads ahead of the sums hipcc kept all 48 values live and spilled 1 (f16) / 7 (bf16) VGPRs
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] += acc1[r];
      asm volatile("" : "+v"(acc0));
      if ((int)gp < M) {
        f32x4v pe[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pe[g] = *reinterpret_cast<const f32x4v*>(posr + 32 * g);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4v v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc0[4 * g + i] + pe[g][i];
          *reinterpret_cast<f32x4v*>(xr + 32 * g) = v;
        }
      }
      return;
    }
    if constexpr (MODE == 2) {
      // (uniform base in SGPRs + 32-bit lane offset: the stores take the saddr form, no 64-bit address arithmetic per lane)
      // The hand-over copy is f16 whatever the operand type T (the affinity build's operands are f16: dss_affinity_f16_u16).
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      typedef typename vec4<f16>::type H4;
      typedef typename vec8<f16>::type H8;
      const unsigned half2 = 64u * (c & 1);
      unsigned char* const k32c = reinterpret_cast<unsigned char*>(kf.k32) + (size_t)(c * LBN * 4);      // uniform
      if constexpr (RT == 1) {                                 // one row tile: the two accumulators are its even / odd k-steps
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] += acc1[r];
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int orow = kf_row(t, eli);
        const unsigned koff = (unsigned)orow * (unsigned)(LK * 4) + 16u * ehh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4v v = t == 0 ? f32x4v{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]}
                                  : f32x4v{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
          if (orow >= 0) *reinterpret_cast<f32x4v*>(k32c + koff + 32 * g) = v;
          H4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            o[i] = from_f32<f16>(v[i]);
            const float rr = to_f32<f16>(o[i]);
            kss[t] = fmaf(rr, rr, kss[t]);
          }
          *reinterpret_cast<H4*>(stg_w + ((half2 + 16 * g) ^ stg_x) + 4096 * t) = o;
          __builtin_amdgcn_sched_barrier(0);                   // one piece at a time: nothing of the next one is started early
        }
      }
      if (!(c & 1)) return;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS write -> read (other lanes' data)
      unsigned char* const k16c = reinterpret_cast<unsigned char*>(C) + (size_t)((c >> 1) * 128);          // uniform
      // rows rq + 8 i of this wave's 32 RT: at most ONE image boundary among them (Tn > 64, host-checked) - one division
      const unsigned gr0 = (unsigned)blockIdx.x * LBM + (unsigned)__builtin_amdgcn_readfirstlane(rloc) + rq;
      const unsigned b0 = __umulhi(gr0, kmagic);
      const unsigned cls0 = b0 * (unsigned)kf.Tn, cls1 = cls0 + (unsigned)kf.Tn;      // the CLS rows of image b0 and b0 + 1
      const unsigned off0 = (gr0 - b0 - 1u) * (unsigned)(LK * 2) + 16u * (unsigned)pq;
#pragma unroll
      for (int i = 0; i < Cfg::NSTORE; ++i) {
        const unsigned gr = gr0 + 8 * i;
        const unsigned off = off0 + (unsigned)(8 * i * LK * 2) - (gr >= cls1 ? (unsigned)(LK * 2) : 0u);
        if ((int)gr < M && gr != cls0 && gr != cls1)
          *reinterpret_cast<H8*>(k16c + off) = *reinterpret_cast<const H8*>(stg + (stg_ro ^ (64u * (i & 1))) + 1024 * i);
      }
      return;
    }
    const unsigned half = 64u * (c & 1);                   // which half of the 128-byte row
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      unsigned char* wp = stg_w + ((half + 16 * g) ^ stg_x);
      if constexpr (GELU == 2) {                           // GELU on packed f16 (kres.h): round first, then 5.5 instructions per value
        static_assert(std::is_same<T, f16>::value, "gelu = 2 is the packed-f16 form: f16 operands only");
        if (RT == 2) {
          h2 p[4] = {{(f16)acc0[4 * g], (f16)acc0[4 * g + 1]}, {(f16)acc0[4 * g + 2], (f16)acc0[4 * g + 3]},
                     {(f16)acc1[4 * g], (f16)acc1[4 * g + 1]}, {(f16)acc1[4 * g + 2], (f16)acc1[4 * g + 3]}};
          gelu_poly_f16xn<4>(p);
          *reinterpret_cast<V4*>(wp) = V4{p[0][0], p[0][1], p[1][0], p[1][1]};
          *reinterpret_cast<V4*>(wp + 4096) = V4{p[2][0], p[2][1], p[3][0], p[3][1]};
        } else {
          h2 p[2] = {{(f16)(acc0[4 * g] + acc1[4 * g]), (f16)(acc0[4 * g + 1] + acc1[4 * g + 1])},
                     {(f16)(acc0[4 * g + 2] + acc1[4 * g + 2]), (f16)(acc0[4 * g + 3] + acc1[4 * g + 3])}};
          gelu_poly_f16xn<2>(p);
          *reinterpret_cast<V4*>(wp) = V4{p[0][0], p[0][1], p[1][0], p[1][1]};
        }
      } else if (RT == 2) {
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
  if constexpr (MODE == 2) kf_finish();
  DSS_TL_FLUSH
}

template <class T, int GELU, int KS, int RT, int NW, int LNM, int MODE = 0>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void linear_kres_kernel(const T* __restrict__ A, float* __restrict__ X,
                                                                 const T* __restrict__ R, long r_ld, long r_plane, float eps,
                                                                 const T* __restrict__ W,
                                                                 const T* __restrict__ bias, const float* __restrict__ aux,
                                                                 T* __restrict__ C, int M, int N, int planar) {
  linear_kres_body<T, GELU, KS, RT, NW, LNM, MODE>(A, X, R, r_ld, r_plane, eps, W, bias, aux, C, M, N, planar, KfOut{nullptr, nullptr, 0, 0.f, nullptr, nullptr, 0, 0, 0});
}

// The K = 768 kernel as the patch embedding (mode 4): M = B Np patches, N = D.
template <class T>
__global__ __launch_bounds__(512, 1) void patch_embed_kres_kernel(const unsigned char* __restrict__ img, const T* __restrict__ Wp,
                                                                  const T* __restrict__ biasp, const float* __restrict__ pos,
                                                                  float* __restrict__ x, int M, int N, int Np, int H, int W, int Wpat) {
  linear_kres_body<T, 0, 48, 1, 8, 0, 4>(nullptr, nullptr, nullptr, 0, 0, 0.f, Wp, biasp, nullptr, nullptr, M, N, 0,
                                             KfOut{x, nullptr, Np, 0.f, img, pos, H, W, Wpat});
}

// The kernel in its hand-over mode (see KfOut): N = K output columns, no bias pointer (it rides in aux); k16 is f16 for both T.
template <class T, int KS, int RT, int NW, int LNM>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void kfeat_kres_kernel(float* __restrict__ X, const T* __restrict__ R, long r_ld, long r_plane, float eps,
                                                            const T* __restrict__ W, const float* __restrict__ aux, f16* __restrict__ k16,
                                                            float* __restrict__ k32, float* __restrict__ rnorm, int M, int Tn, float norm_eps) {
  linear_kres_body<T, 0, KS, RT, NW, LNM, 2>(nullptr, X, R, r_ld, r_plane, eps, W, nullptr, aux, reinterpret_cast<T*>(k16), M, 16 * KS, 0,
                                                 KfOut{k32, rnorm, Tn, norm_eps, nullptr, nullptr, 0, 0, 0});
}

// One wave per output column: Wg[n][k] = T(W[n][k] gamma[k]);  aux[n] = (-sum_k float(Wg[n][k]), bias[n] + sum_k W[n][k] beta[k])
// (sums in fp64: the table is built once per model).
template <class T>
__global__ __launch_bounds__(64) void lnlinear_prepare_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              T* __restrict__ Wg, float* __restrict__ aux, int K) {
  const int n = blockIdx.x, lane = threadIdx.x;
  double sw = 0.0, sb = 0.0;
  for (int k = lane; k < K; k += 64) {
    const float w = W[(size_t)n * K + k];
    const T wg = from_f32<T>(w * gamma[k]);
    Wg[(size_t)n * K + k] = wg;
    sw += (double)to_f32<T>(wg);
    sb += (double)w * (double)beta[k];
  }
  sw = wave_sum(sw);
  sb = wave_sum(sb);
  if (lane == 0) {
    aux[2 * n] = (float)(-sw);
    aux[2 * n + 1] = (float)((double)bias[n] + sb);
  }
}

template <class T, int KS, int RT, int NW, int LNM, int MODE = 0>
static void launch_linear_kres(const void* A, float* X, const void* R, long r_ld, long r_plane, float eps, const void* W,
                               const void* bias, const float* aux, void* C, int M, int N, int gelu, int planar, hipStream_t s) {
  const int blocks = ceil_div(M, LinCfg<KS, RT, NW>::ROWS);
  if constexpr (std::is_same<T, f16>::value) {
    if (gelu == 2) {
      hipLaunchKernelGGL((linear_kres_kernel<T, 2, KS, RT, NW, LNM, MODE>), dim3(blocks), dim3(64 * NW), 0, s, (const T*)A, X,
                         (const T*)R, r_ld, r_plane, eps, (const T*)W, (const T*)bias, aux, (T*)C, M, N, planar);
      return;
    }
  }
  if (gelu)
    hipLaunchKernelGGL((linear_kres_kernel<T, 1, KS, RT, NW, LNM, MODE>), dim3(blocks), dim3(64 * NW), 0, s, (const T*)A, X,
                       (const T*)R, r_ld, r_plane, eps, (const T*)W, (const T*)bias, aux, (T*)C, M, N, planar);
  else
    hipLaunchKernelGGL((linear_kres_kernel<T, 0, KS, RT, NW, LNM, MODE>), dim3(blocks), dim3(64 * NW), 0, s, (const T*)A, X,
                       (const T*)R, r_ld, r_plane, eps, (const T*)W, (const T*)bias, aux, (T*)C, M, N, planar);
}

// A != null: plain Linear.  A == null: LayerNorm prologue on X (res may be null).
template <int KS, int RT, int NW>
static int linear_kres(const char* name, const void* A, float* X, const void* res, int res_layout, float eps, const void* W,
                       const void* bias, const float* aux, void* C, int M, int N, int gelu, int out_layout, int dtype,
                       void* stream) {
  typedef LinCfg<KS, RT, NW> Cfg;
  DSS_REQUIRE((A ? bias != nullptr : (X && aux)) && W && C, "%s: null pointer", name);
  DSS_REQUIRE(gelu == 0 || gelu == 1 || (gelu == 2 && dtype == DSS_F16), "%s: gelu must be 0, 1 (fp32 erf form) or 2 (packed-f16 form, DSS_F16 only) (got %d)", name, gelu);
  DSS_REQUIRE(M > 0 && N > 0 && N % (2 * LBN) == 0 && N <= Cfg::MAXN, "%s: need M > 0, N %% %d == 0, N <= %d (M=%d N=%d)",
              name, 2 * LBN, Cfg::MAXN, M, N);
  DSS_REQUIRE(out_layout == DSS_ROW_MAJOR || out_layout == DSS_PLANAR64,
              "%s: out_layout must be DSS_ROW_MAJOR or DSS_PLANAR64 (got %d)", name, out_layout);
  DSS_REQUIRE(A || res_layout == DSS_ROW_MAJOR || res_layout == DSS_PLANAR64,
              "%s: res_layout must be DSS_ROW_MAJOR or DSS_PLANAR64 (got %d)", name, res_layout);
  DSS_REQUIRE(A || (eps >= 0.f && (const void*)X != res && (const void*)X != C && res != C), "%s: eps < 0 or aliased buffers", name);
  hipStream_t s = (hipStream_t)stream;
  const int planar = out_layout == DSS_PLANAR64;
  const long r_ld = res_layout == DSS_PLANAR64 ? 64 : Cfg::K, r_plane = res_layout == DSS_PLANAR64 ? 64L * M : 64;
#define DSS_LAUNCH_KRES(TT)                                                                                              \
  do {                                                                                                                   \
    if (A) launch_linear_kres<TT, KS, RT, NW, 0>(A, nullptr, nullptr, 0, 0, 0.f, W, bias, nullptr, C, M, N, gelu, planar, s); \
    else if (!res) launch_linear_kres<TT, KS, RT, NW, 1>(nullptr, X, nullptr, 0, 0, eps, W, nullptr, aux, C, M, N, gelu, planar, s); \
    else launch_linear_kres<TT, KS, RT, NW, 2>(nullptr, X, res, r_ld, r_plane, eps, W, nullptr, aux, C, M, N, gelu, planar, s); \
  } while (0)
  switch (dtype) {
    case DSS_F16: DSS_LAUNCH_KRES(f16); break;
    case DSS_BF16: DSS_LAUNCH_KRES(bf16); break;
    default: return fail(DSS_ERR_BAD_ARG, "%s: dtype must be DSS_F16 or DSS_BF16 (got %d)", name, dtype);
  }
#undef DSS_LAUNCH_KRES
  DSS_CHECK_LAUNCH(name);
  return DSS_OK;
}

}  // namespace dss

extern "C" int dss_linear_k384(const void* A, const void* W, const void* bias, void* C, int M, int N, int gelu,
                               int out_layout, int dtype, void* stream) {
  DSS_REQUIRE(A, "dss_linear_k384: null pointer");
  return dss::linear_kres<24, 2, 4>("dss_linear_k384", A, nullptr, nullptr, 0, 0.f, W, bias, nullptr, C, M, N, gelu, out_layout, dtype, stream);
}

extern "C" int dss_linear_k768(const void* A, const void* W, const void* bias, void* C, int M, int N, int gelu,
                               int out_layout, int dtype, void* stream) {
  DSS_REQUIRE(A, "dss_linear_k768: null pointer");
  return dss::linear_kres<48, 1, 8>("dss_linear_k768", A, nullptr, nullptr, 0, 0.f, W, bias, nullptr, C, M, N, gelu, out_layout, dtype, stream);
}


extern "C" int dss_lnlinear_prepare(const float* W, const float* bias, const float* gamma, const float* beta, void* Wg, float* aux,
                                    int N, int K, int dtype, void* stream) {
  DSS_REQUIRE(W && bias && gamma && beta && Wg && aux, "dss_lnlinear_prepare: null pointer");
  DSS_REQUIRE(N > 0 && K > 0, "dss_lnlinear_prepare: need N > 0, K > 0 (N=%d K=%d)", N, K);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case DSS_F16:
      hipLaunchKernelGGL((dss::lnlinear_prepare_kernel<dss::f16>), dim3(N), dim3(64), 0, s, W, bias, gamma, beta, (dss::f16*)Wg, aux, K);
      break;
    case DSS_BF16:
      hipLaunchKernelGGL((dss::lnlinear_prepare_kernel<dss::bf16>), dim3(N), dim3(64), 0, s, W, bias, gamma, beta, (dss::bf16*)Wg, aux, K);
      break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_lnlinear_prepare: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  }
  DSS_CHECK_LAUNCH("dss_lnlinear_prepare");
  return DSS_OK;
}

extern "C" int dss_lnlinear_k384(float* x, const void* residual, int res_layout, float eps, const void* Wg, const float* aux,
                                 void* C, int M, int N, int gelu, int out_layout, int dtype, void* stream) {
  DSS_REQUIRE(x, "dss_lnlinear_k384: null pointer");
  return dss::linear_kres<24, 2, 4>("dss_lnlinear_k384", nullptr, x, residual, res_layout, eps, Wg, nullptr, aux, C, M, N, gelu, out_layout, dtype, stream);
}

extern "C" int dss_patch_embed_p16(const uint8_t* img_u8, const void* Wp, const void* biasp, const float* pos, float* x, int B, int H, int W,
                                   int D, int dtype, void* stream) {
  DSS_REQUIRE(img_u8 && Wp && biasp && pos && x, "dss_patch_embed_p16: null pointer");
  DSS_REQUIRE(B > 0 && H >= 16 && W >= 16, "dss_patch_embed_p16: bad shape B=%d H=%d W=%d", B, H, W);
  typedef dss::LinCfg<48, 1, 8> Cfg;
  DSS_REQUIRE(D > 0 && D % (2 * dss::LBN) == 0 && D <= Cfg::MAXN, "dss_patch_embed_p16: need D %% %d == 0, D <= %d (D=%d)", 2 * dss::LBN, Cfg::MAXN, D);
  const int Hp = H / 16, Wpat = W / 16, Np = Hp * Wpat;
  const long M = (long)B * Np;
  DSS_REQUIRE(M * Np < 4294967296L && (M + B) * D * 4 < 4294967296L && (long)H * W * 3 * B < (1L << 46),
              "dss_patch_embed_p16: B=%d x %d patches exceeds the 32-bit row arithmetic of the epilogue", B, Np);
  const int blocks = dss::ceil_div((int)M, Cfg::ROWS);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case DSS_F16:
      hipLaunchKernelGGL((dss::patch_embed_kres_kernel<dss::f16>), dim3(blocks), dim3(512), 0, s, img_u8, (const dss::f16*)Wp,
                         (const dss::f16*)biasp, pos, x, (int)M, D, Np, H, W, Wpat);
      break;
    case DSS_BF16:
      hipLaunchKernelGGL((dss::patch_embed_kres_kernel<dss::bf16>), dim3(blocks), dim3(512), 0, s, img_u8, (const dss::bf16*)Wp,
                         (const dss::bf16*)biasp, pos, x, (int)M, D, Np, H, W, Wpat);
      break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_patch_embed_p16: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  }
  DSS_CHECK_LAUNCH("dss_patch_embed_p16");
  return DSS_OK;
}

namespace dss {
template <class T, int KS, int RT, int NW>
static void launch_kfeat(float* x, const void* residual, long r_ld, long r_plane, float eps, const void* Wg, const float* aux, float* k32,
                         void* k16, float* rnorm, int M, int T_, float norm_eps, hipStream_t s) {
  const int blocks = ceil_div(M, LinCfg<KS, RT, NW>::ROWS);
  if (residual)
    hipLaunchKernelGGL((kfeat_kres_kernel<T, KS, RT, NW, 2>), dim3(blocks), dim3(64 * NW), 0, s, x, (const T*)residual, r_ld, r_plane, eps,
                       (const T*)Wg, aux, (f16*)k16, k32, rnorm, M, T_, norm_eps);
  else
    hipLaunchKernelGGL((kfeat_kres_kernel<T, KS, RT, NW, 1>), dim3(blocks), dim3(64 * NW), 0, s, x, (const T*)nullptr, r_ld, r_plane, eps,
                       (const T*)Wg, aux, (f16*)k16, k32, rnorm, M, T_, norm_eps);
}
}  // namespace dss

extern "C" int dss_lnlinear_kfeatures(float* x, const void* residual, int res_layout, float eps, const void* Wg, const float* aux,
                                      float* k32, void* k16, float* rnorm, int M, int T, int D, float norm_eps, int dtype, void* stream) {
  DSS_REQUIRE(x && Wg && aux && k32 && k16 && rnorm, "dss_lnlinear_kfeatures: null pointer");
  DSS_REQUIRE(D == 384 || D == 768, "dss_lnlinear_kfeatures: D must be 384 or 768 (got %d)", D);
  DSS_REQUIRE(M > 0 && T > 64 && M % T == 0, "dss_lnlinear_kfeatures: need M = B * T token rows, T > 64 (M=%d T=%d)", M, T);
  DSS_REQUIRE((long)M * T < 4294967296L && (long)M * D * 4 < 4294967296L,
              "dss_lnlinear_kfeatures: M=%d x T=%d exceeds the 32-bit row arithmetic of the hand-over epilogue", M, T);
  DSS_REQUIRE(!residual || res_layout == DSS_ROW_MAJOR || res_layout == DSS_PLANAR64,
              "dss_lnlinear_kfeatures: res_layout must be DSS_ROW_MAJOR or DSS_PLANAR64 (got %d)", res_layout);
  DSS_REQUIRE(eps >= 0.f && norm_eps >= 0.f && (const void*)x != residual && (void*)x != (void*)k32, "dss_lnlinear_kfeatures: eps < 0 or aliased buffers");
  const long r_ld = res_layout == DSS_PLANAR64 ? 64 : D, r_plane = res_layout == DSS_PLANAR64 ? 64L * M : 64;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DSS_F16 && D == 384) dss::launch_kfeat<dss::f16, 24, 2, 4>(x, residual, r_ld, r_plane, eps, Wg, aux, k32, k16, rnorm, M, T, norm_eps, s);
  else if (dtype == DSS_F16) dss::launch_kfeat<dss::f16, 48, 1, 8>(x, residual, r_ld, r_plane, eps, Wg, aux, k32, k16, rnorm, M, T, norm_eps, s);
  else if (dtype == DSS_BF16 && D == 384) dss::launch_kfeat<dss::bf16, 24, 2, 4>(x, residual, r_ld, r_plane, eps, Wg, aux, k32, k16, rnorm, M, T, norm_eps, s);
  else if (dtype == DSS_BF16) dss::launch_kfeat<dss::bf16, 48, 1, 8>(x, residual, r_ld, r_plane, eps, Wg, aux, k32, k16, rnorm, M, T, norm_eps, s);
  else return dss::fail(DSS_ERR_BAD_ARG, "dss_lnlinear_kfeatures: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  DSS_CHECK_LAUNCH("dss_lnlinear_kfeatures");
  return DSS_OK;
}

extern "C" int dss_lnlinear_kfeatures_k384(float* x, const void* residual, int res_layout, float eps, const void* Wg, const float* aux,
                                           float* k32, void* k16, float* rnorm, int M, int T, float norm_eps, void* stream) {
  return dss_lnlinear_kfeatures(x, residual, res_layout, eps, Wg, aux, k32, k16, rnorm, M, T, 384, norm_eps, DSS_F16, stream);
}

extern "C" int dss_lnlinear_k768(float* x, const void* residual, int res_layout, float eps, const void* Wg, const float* aux,
                                 void* C, int M, int N, int gelu, int out_layout, int dtype, void* stream) {
  DSS_REQUIRE(x, "dss_lnlinear_k768: null pointer");
  return dss::linear_kres<48, 1, 8>("dss_lnlinear_k768", nullptr, x, residual, res_layout, eps, Wg, nullptr, aux, C, M, N, gelu, out_layout, dtype, stream);
}



