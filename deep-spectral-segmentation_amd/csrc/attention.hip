// attention.hip - fused multi-head self-attention for the DINO ViT blocks (head dim 64), gfx950 MFMA.
//
// Replaces DINO's Attention.forward after the qkv Linear (reached from
// extract/extract.py:94):   attn = softmax((q @ k^T) * scale) ; x = (attn @ v).transpose(1,2).reshape(B,T,C)
// The [h, T, T] score matrix is never materialised (flash-style online softmax).
//
// One kernel, attn_fwd4 (design notes and the measurements behind them at its definition): the qkv tensor is read in
// place - interleaved [B,T,3,h,64] or DSS_PLANAR64 -, K/V tiles of 64 keys are staged through registers into
// double-buffered LDS, and every wave runs, per 32 keys,
//        S^T[key][q]  = mfma_32x32x16( K-fragment , Q-fragment )      (contraction over dh = 64)
//        online softmax down each lane's own query column (lane-local: no cross-lane traffic on the common path)
//        O^T[dh][q]  += mfma_32x32x16( V^T-fragment , P^T-fragment )  (contraction over 32 keys)
// Computing the TRANSPOSED score tile makes the softmax reduction lane-local and lets the fp32 probabilities be
// packed straight into the B operand of the second MFMA - no LDS round trip.
//
// MFMA layouts used (v_mfma_f32_32x32x16_{f16,bf16}; cdna_hip_programming.md §3):
//   A operand: lane l holds A[i = l&31][k = 8*(l>>5) + e], e = 0..7   (8 halves = 16 B)
//   B operand: lane l holds B[k = 8*(l>>5) + e][j = l&31]
//   C/D      : lane l, register r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
// Only the (i, j) maps matter for correctness: the k index is summed, so any bijection of k is valid as
// long as A and B use the same one.
//
// KEY ORDER.  After the first MFMA lane l (half hh = l>>5) holds, for its query, keys
//   key(r) = (r&3) + 8*(r>>2) + 4*hh,  r = 0..15  of the 32-key block.  Registers 8t..8t+7 (t = 0,1) form
// the B operand of P.V MFMA number t, i.e. operand slot (hh, e) carries key 16t + 8*(e>>2) + 4*hh + (e&3); the
// transposed LDS read of V (ds_read_b64_tr_b16) delivers V^T fragments in exactly that key order.
#include "common.h"

// scripts/probes/attn_clock_probe.hip includes this file with DSS_ATTN_CLOCK defined to read the shader clock the chip
// sustains INSIDE the kernel, scripts/probes/attn_timeline_probe.hip with DSS_ATTN_TIMELINE to record when and where
// every workgroup ran; in the library the hooks compile to nothing.
#ifdef DSS_ATTN_CLOCK   // shader clock sustained inside the kernel: s_memtime (shader cycles) vs s_memrealtime (100 MHz)
__device__ unsigned long long dss_clock_buf[4];
#define DSS_CLOCK_BEGIN                                                                       \
  const bool clk_on = blockIdx.x == DSS_PROBE_BLOCK && threadIdx.x == 0;                      \
  unsigned long long clk_c0 = 0, clk_r0 = 0;                                                  \
  if (clk_on) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = wall_clock64(); }
#define DSS_CLOCK_END                                                                         \
  if (clk_on) {                                                                               \
    dss_clock_buf[0] = __builtin_readcyclecounter() - clk_c0;                                 \
    dss_clock_buf[1] = wall_clock64() - clk_r0;                                               \
  }
#elif defined(DSS_ATTN_TIMELINE)   // per-workgroup start / end (100 MHz counter) and placement (HW_ID, XCC_ID)
__device__ unsigned long long* dss_timeline_buf;   // [gridDim.x][4]
#define DSS_CLOCK_BEGIN                                                                       \
  unsigned long long tl_r0 = 0;                                                               \
  if (threadIdx.x == 0) tl_r0 = wall_clock64();
#define DSS_CLOCK_END                                                                         \
  if (threadIdx.x == 0) {                                                                     \
    unsigned hw, xcc;                                                                         \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                          \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                        \
    unsigned long long* t = dss_timeline_buf + 4ull * blockIdx.x;                             \
    t[0] = tl_r0; t[1] = wall_clock64(); t[2] = hw; t[3] = xcc;                               \
  }
#else
#define DSS_CLOCK_BEGIN
#define DSS_CLOCK_END
#endif

namespace dss {

static constexpr int DH = 64;  // head dim of every DINO ViT

// ---- LDS tile layouts (both conflict-free, checked with SQ_LDS_BANK_CONFLICT) -----------------------------------------
//   * K tile row-major, row stride 144 B: the ds_read_b128 operand reads (16 lanes = 16 rows) land on 16 distinct
//     4-bank slots.
//   * V tile row-major, row stride 192 B, read with ds_read_b64_tr_b16: inside a 16-lane group lane i supplies the
//     address of (row i>>2, cols 4*(i&3)..+3) of a [4 keys x 16 dh] block and receives column i (verified on hardware,
//     scripts/probes/tr16_probe.hip) - the hardware transpose turns row-major V into the V^T fragment the P.V MFMA
//     needs; 192 B puts the 4 rows of a block on disjoint bank quarters.
template <class T>
__device__ __forceinline__ typename vec8<T>::type lds_read_tr_pair(const T* p_lo, const T* p_hi) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p_lo));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p_hi));
  const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(typename vec8<T>::type, c);
}

static constexpr int KLD = 72;   // halves per K row in LDS (144 B)
static constexpr int VLD = 96;   // halves per V row in LDS (192 B)

// Cross-half (lane ^ 32) exchange on the VALU (v_permlane32_swap), no LDS round trip: returns, in every lane,
// max(x[lane & 31], x[32 + (lane & 31)]).
// NOTE (hipcc / ROCm 7.2 front-end bug): __builtin_bit_cast(float, r[1]) applied directly to an element of the
// builtin's 2-vector result reads element 0 (seen in the -O0 IR: both loads use the vector's base address), which
// silently turned max(r0, r1) into r0 and r0 + r1 into 2*r0.  Copy the elements into scalars first.
__device__ __forceinline__ float half_pair_max(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned lo = r[0], hi = r[1];
  return fmaxf(__uint_as_float(lo), __uint_as_float(hi));
}
__device__ __forceinline__ float half_pair_sum(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned lo = r[0], hi = r[1];
  return __uint_as_float(lo) + __uint_as_float(hi);
}

// ================================================================================================
// attn_fwd4: occupancy instead of choreography.  What round 2 measured on MI355X (scripts/probes/overlap_probe.hip,
// profiles/r02_attention_probes.txt) and what it means at head dim 64, where the online softmax costs ~45 VALU
// instructions per 8 MFMAs:
//   * ONE wave issues at most one VALU instruction every ~5-7 cycles (a v_exp_f32 every ~9), whatever else the SIMD does;
//     a SIMD reaches its VALU rate (2.7 cycles per simple op, 4.7 per v_exp, this mix ~3.2) only with 3-4 resident waves;
//   * the matrix pipe is indifferent to VALU traffic from OTHER waves (16 MFMAs: 549 cycles alone, 553 beside a softmax);
//   * the loop body below on registers only runs 509 / 385 / 334 / 307 cycles per 8 MFMAs (256 cycles of matrix pipe) per
//     SIMD at 1 / 2 / 3 / 4 waves per SIMD.
// Two kernels built on two waves per SIMD lost to that arithmetic and were deleted: round 1's 4-wave x 64-query kernel
// (182 VGPRs; 566-600 us on the 290-image bench shape) and an 8-wave ping-pong kernel that kept one wave of every SIMD in
// a pure-MFMA phase while its partner ran the softmax (223-250 VGPRs, 4 barriers per key tile: 680-740 us - the single
// softmax wave per SIMD issues too slowly to keep up with the matrix phase, so the phases added up instead of hiding
// each other).  Here a wave owns 32 queries instead of 64 - O^T is 32 registers, the kernel fits 128 - so FOUR waves
// share a SIMD (two 8-wave workgroups per CU) and the hardware interleaves one wave's MFMAs with the others' softmax:
// 541 us, 0.67 PFLOP/s.  The price: every K / V^T fragment read from LDS feeds one MFMA instead of two.
//   * workgroup = 8 waves x 32 queries = 256 queries of one (image, head), 1-D XCD-aware grid (the query blocks of one
//     (image, head) share an XCD, so K/V come out of its L2: PMC traffic == algorithmic bytes); K/V tiles of 64 keys are
//     staged once per workgroup through registers into double-buffered LDS, one barrier per tile;
//   * per 32 keys: S^T = K.Q^T (4 MFMAs from the inline constant 0), VALU-diet softmax IN PLACE - 8 v_pk_fma_f32 +
//     16 v_exp_f32 + 9 v_pk_add_f32 + 8 v_cvt_pk: the probabilities are computed against the OLD running max straight
//     away and their row sums are the rescale test (every p <= 2^6 is implied by sum(p) <= 2^6; a first tile, m = -1e30,
//     or an overflow gives inf/NaN, which fails `sum <= 2^6` as well) - no max tree, no cross-lane exchange; then
//     O^T += V^T.P^T (4 MFMAs);
//   * when the test fires (wave-uniform ballot; first tile, or a row maximum that grew by more than 2^6) the raw scores
//     are recomputed by re-issuing the 4 MFMAs - their registers hold the probabilities by then, and spare registers for
//     a copy would cost the fourth wave - and take the exact path: new max, rescale of O and l, probabilities again;
//   * the ragged last key tile is peeled out of the main loop (instruction issue is the scarce resource: the main loop
//     carries no per-half conditions); its keys past the end are masked through the MFMA accumulator's INITIAL value
//     (-inf in their rows, 0 elsewhere - the MFMA adds it for free).
// Where the rest goes (profiles/r02_attention_probes.txt): the register-only body runs 307 cycles per half per SIMD, the
// kernel 578.  Ablations of this kernel: fragments from registers instead of LDS -4 %, no restaging -4 %, no restaging
// and no barrier -9 %; the timeline probe shows 1.61 of the 2 workgroup slots of a CU occupied on average - 6.5 us pass
// between the end of a workgroup (its output stores drain before its registers and LDS are released) and the start of
// its successor, 16 % of a 40 us workgroup.  Tried against that and measured slower: workgroups that walk several work
// items (2 / 4 / 8 / all of a CU slot's ~14: +1 ... +20 %; item times vary 23-74 us with what the neighbours are doing, so
// static walks lose to the hardware dispatcher what they save on the gap), a PERSISTENT grid of two workgroups per CU
// that draw items dynamically from per-XCD ticket counters, the next ticket fetched while the current item runs
// (bit-identical results; 540 vs 535 us at T = 901, 105 vs 49 us at T = 197 - the "gap" is the item's own prologue and
// drain, not dispatch latency, and only a second resident workgroup hides it), a three-tile LDS ring with the score
// MFMAs of the next half issued ahead of this half's softmax (128 VGPRs + spills: 3x slower), and the same software
// pipeline done properly at two waves per SIMD - 64 queries per wave as two independent 32-query tiles A / B, per 32 keys
// the steps {S_B = K.Q_B, O_B += V.P_B || softmax A} and {O_A += V.P_A, S_A(next) = K.Q_A || softmax B} with the
// fragment reads of the next step issued first, every K / V^T fragment feeding two MFMAs, a 3-deep LDS ring with one
// mid-tile barrier; 250 VGPRs, no spills, hipcc interleaves each step's 8 MFMAs with its ~45 VALU instructions
// (sched_group_barrier), results bit-identical to this kernel: 588 vs 571 us on the same box.  Neither pipe is
// saturated in either kernel (a SIMD retires one instruction per ~10 cycles); what is left needs instruction-level
// control of issue and dependency stalls that the compiler does not give.
template <class T>
__global__ __launch_bounds__(512, 4) void attn_fwd4_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn,
                                                           int heads, int nb, int nqb, float scale_log2,
                                                           int planar) {
  typedef typename vec8<T>::type V8;
  typedef typename vec4<T>::type V4;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) T Ks[2][64 * KLD];
  __shared__ __attribute__((aligned(16))) T Vs[2][64 * VLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  int qblk, group;
  // XCD-aware block order (1-D grid).  Workgroup id -> XCD is observed round-robin (id % 8) and every XCD has its own
  // L2: the nqb query blocks of one (image, head) are given ids that are congruent mod 8, so they run on ONE XCD close
  // in time and its L2 serves K/V to all of them.  Pure speed heuristic - any placement is correct.
  {
    const int id = blockIdx.x, groups = heads * nb, g8 = groups & ~7;
    if (id < nqb * g8) {
      const int xcd = id & 7, slot = id >> 3;
      group = (slot / nqb) * 8 + xcd;
      qblk = slot % nqb;
    } else {
      const int r = id - nqb * g8;
      group = g8 + r / nqb;
      qblk = r % nqb;
    }
  }
  const int head = group % heads, b = group / heads;
  const long plane = (long)nb * Tn * DH;
  const long rs = planar ? DH : 3L * heads * DH;                    // row stride (halves)
  const long koff = planar ? heads * plane : (long)heads * DH;      // q -> k ; q -> v is twice that
  const T* base = planar ? qkv + head * plane + (long)b * Tn * DH : qkv + (long)b * Tn * rs + (long)head * DH;
  const int q0 = qblk * 256 + wave * 32;
  const bool active = __builtin_amdgcn_readfirstlane((int)(q0 < Tn)) != 0;
  DSS_CLOCK_BEGIN

  V8 qf[4];
  {
    int qa = q0 + li;
    qa = qa < Tn ? qa : Tn - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const V8*>(base + (long)qa * rs + 16 * s + 8 * hh);
  }
  const int srow = tid >> 3, scol = (tid & 7) * 8;
  V8 kreg, vreg;
  auto stage_load = [&](int kt) {
    const int key = kt * 64 + srow;
    if (key < Tn) {
      const T* p = base + (long)key * rs + scol;
      kreg = *reinterpret_cast<const V8*>(p + koff);
      vreg = *reinterpret_cast<const V8*>(p + 2 * koff);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { kreg[i] = (T)0.f; vreg[i] = (T)0.f; }
    }
  };
  auto stage_write = [&](int buf) {
    *reinterpret_cast<V8*>(&Ks[buf][srow * KLD + scol]) = kreg;
    *reinterpret_cast<V8*>(&Vs[buf][srow * VLD + scol]) = vreg;
  };

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -1.0e30f, mc = -1.0e30f * scale_log2;
  f32x2 l2 = {0.f, 0.f};                   // this lane's partial row sums (two interleaved halves of its keys)
  const float c = scale_log2;
  const int nkt = (Tn + 63) / 64;
  const int tr_row = 4 * hh + ((lane & 15) >> 2);
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  // one 32-key half: scores, diet softmax in place, P.V.  `tail`: keys past the end of the sequence are masked through
  // the accumulator's initial value (last half only); full halves start the MFMA chain from the inline constant 0.
  // (Two separate chains on purpose: one chain behind a select would materialise sixteen zeros per half.)
  auto half_block = [&](const T* kbuf, const T* vbuf, int half, int key0, bool tail) {
    const T* krow = kbuf + (half * 32 + li) * KLD + 8 * hh;
    auto scores = [&]() {
      f32x16 s;
      if (tail) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = (key0 + (r & 3) + 8 * (r >> 2) + 4 * hh) < Tn ? 0.f : -INFINITY;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) s = mfma32x32x16(*reinterpret_cast<const V8*>(krow + 16 * sl), qf[sl], s);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) s = mfma32x32x16(*reinterpret_cast<const V8*>(krow + 16 * sl), qf[sl], s);
      }
      return s;
    };
    f32x16 s = scores();
    // ---- diet softmax, in place: s becomes p = exp2(s c - m c) against the OLD running max ----
    const f32x2 c2 = {c, c};
    f32x2 nmc2 = {-mc, -mc}, acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x2 sv = {s[2 * i], s[2 * i + 1]};
      const f32x2 e = __builtin_elementwise_fma(sv, c2, nmc2);
      s[2 * i] = __builtin_amdgcn_exp2f(e[0]);
      s[2 * i + 1] = __builtin_amdgcn_exp2f(e[1]);
    }
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      acc0 += f32x2{s[2 * i], s[2 * i + 1]};
      acc1 += f32x2{s[2 * i + 2], s[2 * i + 3]};
    }
    acc0 += acc1;
    // every p <= 2^6 is implied by both partial sums <= 2^6; inf / NaN (first tile: m = -1e30) fail the test too
    if (__builtin_amdgcn_ballot_w64(!(acc0[0] <= 64.0f && acc0[1] <= 64.0f)) != 0) {   // wave-uniform, rare
      s = scores();
      float mx = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])),
                           fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]))));
      mx = half_pair_max(mx);
      const float m_new = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);   // m = -1e30 initially -> alpha = 0
      m = m_new;
      mc = m_new * c;
      l2 *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      acc0 = f32x2{0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -mc));
        s[r + 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r + 1], c, -mc));
        acc0 += f32x2{s[r], s[r + 1]};
      }
    }
    l2 += acc0;
    V8 pb0, pb1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pb0[e] = from_f32<T>(s[e]);
      pb1[e] = from_f32<T>(s[8 + e]);
    }
    const T* vbase = vbuf + (half * 32 + tr_row) * VLD + tr_col;
    {
      const V8 v0 = lds_read_tr_pair<T>(vbase, vbase + 8 * VLD);
      const V8 v1 = lds_read_tr_pair<T>(vbase + 32, vbase + 8 * VLD + 32);
      o0 = mfma32x32x16(v0, pb0, o0);
      o1 = mfma32x32x16(v1, pb0, o1);
    }
    {
      const V8 v0 = lds_read_tr_pair<T>(vbase + 16 * VLD, vbase + 24 * VLD);
      const V8 v1 = lds_read_tr_pair<T>(vbase + 16 * VLD + 32, vbase + 24 * VLD + 32);
      o0 = mfma32x32x16(v0, pb1, o0);
      o1 = mfma32x32x16(v1, pb1, o1);
    }
  };

  stage_load(0);
  stage_write(0);
  __syncthreads();
  // full tiles: straight-line code, no per-half conditions (instruction issue is the scarce resource of this kernel)
  const int nfull = Tn / 64;
  int kt = 0;
  for (; kt < nfull; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) stage_load(kt + 1);
    if (active) {
      half_block(&Ks[buf][0], &Vs[buf][0], 0, kt * 64, false);
      half_block(&Ks[buf][0], &Vs[buf][0], 1, kt * 64 + 32, false);
    }
    if (kt + 1 < nkt) stage_write(buf ^ 1);
    __syncthreads();
  }
  if (kt < nkt && active) {                    // the ragged last tile: 1..63 real keys
    const int buf = kt & 1, key0 = kt * 64;
    half_block(&Ks[buf][0], &Vs[buf][0], 0, key0, key0 + 32 > Tn);
    if (key0 + 32 < Tn) half_block(&Ks[buf][0], &Vs[buf][0], 1, key0 + 32, true);
  }
  float l = l2[0] + l2[1];
  DSS_CLOCK_END

  if (!active) return;
  const float lsum = half_pair_sum(l);                  // the two half-waves hold disjoint keys of each query
  const int q = q0 + li;
  if (q < Tn) {
    const float inv = 1.0f / lsum;
    T* orow = out + ((long)b * Tn + q) * heads * DH + (long)head * DH;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      V4 a, cc;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = from_f32<T>(o0[4 * g + i] * inv);
        cc[i] = from_f32<T>(o1[4 * g + i] * inv);
      }
      *reinterpret_cast<V4*>(orow + 8 * g + 4 * hh) = a;
      *reinterpret_cast<V4*>(orow + 32 + 8 * g + 4 * hh) = cc;
    }
  }
}

template <class T>
static void launch_attention(const void* qkv, void* out, int B, int Tn, int heads, float scale, hipStream_t s,
                             int planar) {
  const int nqb = ceil_div(Tn, 256);
  hipLaunchKernelGGL((attn_fwd4_kernel<T>), dim3((unsigned)(nqb * heads * B)), dim3(512), 0, s, (const T*)qkv,
                     (T*)out, Tn, heads, B, nqb, scale * 1.4426950408889634f, planar);
}

}  // namespace dss

extern "C" int dss_attention_fwd(const void* qkv, int qkv_layout, void* out, int B, int T, int heads, float scale,
                                 int dtype, void* stream) {
  DSS_REQUIRE(qkv && out, "dss_attention_fwd: null pointer");
  DSS_REQUIRE(qkv_layout == DSS_ROW_MAJOR || qkv_layout == DSS_PLANAR64,
              "dss_attention_fwd: qkv_layout must be DSS_ROW_MAJOR or DSS_PLANAR64 (got %d)", qkv_layout);
  DSS_REQUIRE(B > 0 && T > 0 && heads > 0, "dss_attention_fwd: bad shape B=%d T=%d heads=%d", B, T, heads);
  DSS_REQUIRE((long)B * heads * dss::ceil_div(T, 256) < 2147483647L, "dss_attention_fwd: too many workgroups");
  hipStream_t s = (hipStream_t)stream;
  const int planar = qkv_layout == DSS_PLANAR64;
  switch (dtype) {
    case DSS_F16: dss::launch_attention<dss::f16>(qkv, out, B, T, heads, scale, s, planar); break;
    case DSS_BF16: dss::launch_attention<dss::bf16>(qkv, out, B, T, heads, scale, s, planar); break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_attention_fwd: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  }
  DSS_CHECK_LAUNCH("attention");
  return DSS_OK;
}
