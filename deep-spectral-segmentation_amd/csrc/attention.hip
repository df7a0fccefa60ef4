// attention.hip - fused multi-head self-attention for the DINO ViT blocks (head dim 64), gfx950 MFMA.
//
// Replaces DINO's Attention.forward after the qkv Linear (reached from
// extract/extract.py:94):   attn = softmax((q @ k^T) * scale) ; x = (attn @ v).transpose(1,2).reshape(B,T,C)
// The [h, T, T] score matrix is never materialised (flash-style online softmax).
//
// Kernels (both read the qkv tensor in place - interleaved [B,T,3,h,64] or DSS_PLANAR64 - and stage K/V tiles of 64
// keys through registers into double-buffered LDS):
//  * attn_fwd2: 4 waves x 64 queries per workgroup, one barrier per key tile; every wave runs
//        S^T[key][q]  = mfma_32x32x16( K-fragment , Q-fragment )      (contraction over dh = 64)
//        online softmax down each lane's own query column (16 registers + one lane^32 exchange)
//        O^T[dh][q]  += mfma_32x32x16( V^T-fragment , P^T-fragment )  (contraction over 32 keys)
//    back to back.  Computing the TRANSPOSED score tile makes the softmax reduction lane-local and lets the fp32
//    probabilities be packed straight into the B operand of the second MFMA - no LDS round trip.
//  * attn_fwd3 (default): the same arithmetic, 8 waves x 64 queries per workgroup in two groups that PING-PONG on
//    every SIMD: while one group is in its matrix phase (P.V of the previous 32 keys + Q.K^T of the next 32: 16 MFMAs,
//    nothing else but LDS fragment reads) the other runs its softmax phase (~140 VALU instructions, no MFMA), then they
//    swap - see the kernel.  At head dim 64 the VALU work of the softmax is as long as the MFMA work; kept in one
//    instruction stream the two serialise, split over the two waves of a SIMD they overlap by construction.
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

namespace dss {

static constexpr int DH = 64;  // head dim of every DINO ViT

// ================================================================================================
// v2: LDS-staged, 64 queries per wave, no pack pass.
//   * block = 4 waves = 256 query rows of one (image, head); K and V tiles of 64 keys are read ONCE per
//     block straight from the qkv tensor (16 B/lane, 128-B row segments), staged through registers into a
//     double-buffered LDS image and shared by the 4 waves (one barrier per tile; the next tile's global
//     loads are issued before the MFMAs of the current one and written to LDS after them).
//   * K tile row-major, row stride 144 B: the ds_read_b128 operand reads (16 lanes = 16 rows) land on 16
//     distinct 4-bank slots.  V tile row-major, row stride 192 B, read with ds_read_b64_tr_b16: inside
//     a 16-lane group lane i supplies the address of (row i>>2, cols 4*(i&3)..+3) of a [4 keys x 16 dh]
//     block and receives column i (verified on hardware, scripts/probes/tr16_probe.hip) - the hardware
//     transpose turns row-major V into the V^T fragment the P.V MFMA needs; 192 B puts the 4 rows of a
//     block on disjoint bank quarters.
//   * each wave holds TWO 32-query blocks, so every K / V^T fragment read from LDS feeds two MFMAs.
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

// Online-softmax update for one 32-query block on a 32-key score tile (raw q.k scores in s).
//   m   : running max in RAW score units (shared by both half-waves of a query)
//   mc  : m * c  (c = scale * log2 e), so p = exp2(fma(s, c, -mc)) is one FMA + one v_exp per score
//   l   : THIS LANE's partial row sum (its half of the keys); the two halves are added once, in the epilogue
// The O accumulators are rescaled only when some query's max grew by more than RESCALE_THR (in log2 units):
// until then p <= 2^RESCALE_THR, harmless for f16/bf16 operands and the f32 accumulators.  The decision is
// wave-uniform (ballot), taken BEFORE this tile's probabilities exist, so every p, l and O stays consistent.
static constexpr float RESCALE_THR = 6.0f;
template <class T>
__device__ __forceinline__ void softmax_block(f32x16& s, float& m, float& mc, float& l, f32x16& o0, f32x16& o1,
                                              typename vec8<T>::type& pb0, typename vec8<T>::type& pb1, float c,
                                              bool tail, int key0, int hh, int Tn) {
  if (tail) {   // (attn_fwd3 passes tail = false: it masks through the MFMA accumulator's initial value instead)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (key >= Tn) s[r] = -INFINITY;
    }
  }
  float mx = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
  mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11])), fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]))));
  mx = half_pair_max(mx);
  if (__builtin_amdgcn_ballot_w64((mx - m) * c > RESCALE_THR) != 0) {  // wave-uniform
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);       // m = -1e30 initially -> alpha = 0
    m = m_new;
    mc = m_new * c;
    l *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
  }
  float rs = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -mc));
    rs += s[r];
  }
  l += rs;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    pb0[e] = from_f32<T>(s[e]);
    pb1[e] = from_f32<T>(s[8 + e]);
  }
}

template <class T>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn,
                                                           int heads, int nb, int nqb, float scale_log2,
                                                           int planar) {
  typedef typename vec8<T>::type V8;
  typedef typename vec4<T>::type V4;
  __shared__ __attribute__((aligned(16))) T Ks[2][64 * KLD];
  __shared__ __attribute__((aligned(16))) T Vs[2][64 * VLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  // XCD-aware block order (1-D grid).  Workgroup id -> XCD is observed round-robin (id % 8) and every XCD has
  // its own L2: the nqb query-blocks of one (image, head) are given ids that are congruent mod 8, so they run
  // on ONE XCD close in time and its L2 serves K/V to all of them (PMC before: K/V fetched nqb x from HBM).
  // Pure speed heuristic - any placement is correct.
  int qblk, group;
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
  // interleaved qkv [B, T, 3, h, 64]: row stride 3*h*64, k at +h*64, v at +2*h*64 from q.
  // planar qkv [3*h][B*T][64] (what dss_linear_k384 writes with DSS_PLANAR64): every (q|k|v, head) is a plane of
  // contiguous 128-byte rows, so the K/V tiles of one (image, head) are contiguous 8 KB runs.
  const long plane = (long)nb * Tn * DH;
  const long rs = planar ? DH : 3L * heads * DH;                    // row stride (halves)
  const long koff = planar ? heads * plane : (long)heads * DH;      // q -> k ; q -> v is twice that
  const T* base = planar ? qkv + head * plane + (long)b * Tn * DH : qkv + (long)b * Tn * rs + (long)head * DH;
  const int q0 = qblk * 256 + wave * 64;
  const bool active = q0 < Tn;                                      // wave-uniform

  // ---- Q fragments (registers, once): lane -> query row q0 + 32*qb + li, dh slice 16*s + 8*hh ----------
  V8 qf0[4], qf1[4];
  {
    int qa = q0 + li, qb = q0 + 32 + li;
    qa = qa < Tn ? qa : Tn - 1;
    qb = qb < Tn ? qb : Tn - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf0[s] = *reinterpret_cast<const V8*>(base + (long)qa * rs + 16 * s + 8 * hh);
      qf1[s] = *reinterpret_cast<const V8*>(base + (long)qb * rs + 16 * s + 8 * hh);
    }
  }
  // ---- staging map: chunk c = tid + 256*j (j = 0,1): row c>>3, 8 halves at column 8*(c&7) ---------------
  const int srow0 = tid >> 3, scol = (tid & 7) * 8;
  V8 kreg[2], vreg[2];
  auto stage_load = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = kt * 64 + srow0 + 32 * j;
      if (key < Tn) {
        const T* p = base + (long)key * rs + scol;
        kreg[j] = *reinterpret_cast<const V8*>(p + koff);
        vreg[j] = *reinterpret_cast<const V8*>(p + 2 * koff);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { kreg[j][i] = (T)0.f; vreg[j][i] = (T)0.f; }
      }
    }
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = srow0 + 32 * j;
      *reinterpret_cast<V8*>(&Ks[buf][r * KLD + scol]) = kreg[j];
      *reinterpret_cast<V8*>(&Vs[buf][r * VLD + scol]) = vreg[j];
    }
  };

  f32x16 oa0, oa1, ob0, ob1;   // O^T accumulators: query block a/b x dh block 0/1
#pragma unroll
  for (int r = 0; r < 16; ++r) { oa0[r] = 0.f; oa1[r] = 0.f; ob0[r] = 0.f; ob1[r] = 0.f; }
  float ma = -1.0e30f, mca = -1.0e30f * scale_log2, la = 0.f, mb = -1.0e30f, mcb = -1.0e30f * scale_log2, lb = 0.f;

  const int nkt = (Tn + 63) / 64;
  stage_load(0);
  stage_write(0);
  __syncthreads();
  // per-lane constants of the transposed V read: 16-lane group g = lane>>4, i = lane&15
  const int tr_row = 4 * hh + ((lane & 15) >> 2);           // + 16*t (+8 for the second half)
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);  // + 32*db
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) stage_load(kt + 1);
    if (active) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int key0 = kt * 64 + half * 32;
        if (key0 < Tn) {
          f32x16 sa, sb;
#pragma unroll
          for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
          const T* krow = &Ks[buf][(half * 32 + li) * KLD + 8 * hh];
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) {
            const V8 kf = *reinterpret_cast<const V8*>(krow + 16 * sl);
            sa = mfma32x32x16(kf, qf0[sl], sa);
            sb = mfma32x32x16(kf, qf1[sl], sb);
          }
          const bool tail = key0 + 32 > Tn;
          V8 pa0, pa1, pb0, pb1;
          softmax_block<T>(sa, ma, mca, la, oa0, oa1, pa0, pa1, scale_log2, tail, key0, hh, Tn);
          softmax_block<T>(sb, mb, mcb, lb, ob0, ob1, pb0, pb1, scale_log2, tail, key0, hh, Tn);
          const T* vbase = &Vs[buf][(half * 32 + tr_row) * VLD + tr_col];
          {  // t = 0 (keys 0..15 of the block), dh blocks 0 and 1
            const V8 v0 = lds_read_tr_pair<T>(vbase, vbase + 8 * VLD);
            const V8 v1 = lds_read_tr_pair<T>(vbase + 32, vbase + 8 * VLD + 32);
            oa0 = mfma32x32x16(v0, pa0, oa0);
            ob0 = mfma32x32x16(v0, pb0, ob0);
            oa1 = mfma32x32x16(v1, pa0, oa1);
            ob1 = mfma32x32x16(v1, pb0, ob1);
          }
          {  // t = 1 (keys 16..31)
            const V8 v0 = lds_read_tr_pair<T>(vbase + 16 * VLD, vbase + 24 * VLD);
            const V8 v1 = lds_read_tr_pair<T>(vbase + 16 * VLD + 32, vbase + 24 * VLD + 32);
            oa0 = mfma32x32x16(v0, pa1, oa0);
            ob0 = mfma32x32x16(v0, pb1, ob0);
            oa1 = mfma32x32x16(v1, pa1, oa1);
            ob1 = mfma32x32x16(v1, pb1, ob1);
          }
        }
      }
    }
    if (kt + 1 < nkt) stage_write(buf ^ 1);
    __syncthreads();
  }

  if (!active) return;
  auto store_q = [&](int q, const f32x16& x0, const f32x16& x1, float l) {
    if (q >= Tn) return;
    const float inv = 1.0f / l;
    T* orow = out + ((long)b * Tn + q) * heads * DH + (long)head * DH;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      V4 a, c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = from_f32<T>(x0[4 * g + i] * inv);
        c[i] = from_f32<T>(x1[4 * g + i] * inv);
      }
      *reinterpret_cast<V4*>(orow + 8 * g + 4 * hh) = a;
      *reinterpret_cast<V4*>(orow + 32 + 8 * g + 4 * hh) = c;
    }
  };
  store_q(q0 + li, oa0, oa1, half_pair_sum(la));       // the two half-waves hold disjoint keys of each query
  store_q(q0 + 32 + li, ob0, ob1, half_pair_sum(lb));
}

// ================================================================================================
// v3: 8 waves in two ping-pong groups (X = waves 0-3, Y = waves 4-7; a workgroup's waves w and w + 4 share a SIMD).
// Time is cut into PHASES separated by raw s_barriers.  In an M phase a wave issues nothing but MFMAs and the LDS
// fragment reads that feed them: P.V of the previous 32-key half (8 MFMAs) and Q.K^T of the next half (8 MFMAs).  In a
// V phase it runs the online softmax of the scores it just produced (~140 VALU instructions, no MFMA).  Group Y runs one
// phase behind group X, so on every SIMD one wave is in M while its partner is in V:
//
//   phase      4t            4t+1            4t+2                 4t+3       (tile t = keys 64t..64t+63, halves 2t / 2t+1)
//   X      M: PV(2t-1)     V: soft(2t)     M: PV(2t), QK(2t+1)  V: soft(2t+1)
//             QK(2t)          frags(2t+1)     + restage            frags(2t+2)
//   Y      V: soft(2t-1)   M: PV(2t-1)     V: soft(2t)          M: PV(2t), QK(2t+1)
//             frags(2t)       QK(2t)          frags(2t+1) + restage
//   frags(h) = the LDS fragment reads of the M phase of half h, issued at the end of the V phase before it;
//   restage  = all waves write tile t+1 (global loads issued a tile earlier) into the other LDS buffer - it held tile
//              t-1, whose last fragment reads (Y's frags(2t)) were issued before the barrier ending phase 4t; tile t+1
//              is first read by X's frags(2t+2) in phase 4t+3 - and issue the global loads of tile t+2.
// After the last tile X finishes with PV of the last half and Y with its softmax + PV.  Every wave passes the same
// number of barriers whatever its queries (inactive waves still stage).
template <class T>
__global__ __launch_bounds__(512, 2) void attn_fwd3_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn,
                                                           int heads, int nb, int nqb, float scale_log2,
                                                           int planar) {
  typedef typename vec8<T>::type V8;
  typedef typename vec4<T>::type V4;
  __shared__ __attribute__((aligned(16))) T Ks[2][64 * KLD];
  __shared__ __attribute__((aligned(16))) T Vs[2][64 * VLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const bool group_x = wave < 4;
  int qblk, group;
  {  // XCD-aware block order: the query blocks of one (image, head) get ids congruent mod 8 (see attn_fwd2)
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
  const int q0 = qblk * 512 + wave * 64;
  // wave-uniform, and provably so for the compiler (scalar branches, no exec masking around the MFMAs).  A wave whose
  // second 32-query block lies past the sequence computes it anyway on clamped rows (at most one wave per image and
  // head; its partner on the SIMD is a full wave) and simply does not store it.
  const bool active = __builtin_amdgcn_readfirstlane((int)(q0 < Tn)) != 0;

  // ---- Q fragments (registers, once) ---------------------------------------------------------------------
  V8 qf0[4], qf1[4];
  {
    int qa = q0 + li, qb = q0 + 32 + li;
    qa = qa < Tn ? qa : Tn - 1;
    qb = qb < Tn ? qb : Tn - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf0[s] = *reinterpret_cast<const V8*>(base + (long)qa * rs + 16 * s + 8 * hh);
      qf1[s] = *reinterpret_cast<const V8*>(base + (long)qb * rs + 16 * s + 8 * hh);
    }
  }
  // ---- staging map: 512 threads x one 16-byte piece of K and of V per tile: row tid>>3, 8 halves at 8*(tid&7) ----
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
  // phase barrier: raw s_barrier (no release fence: the compiler would drain vmcnt - the prefetch of the next tile -
  // in front of every barrier); LDS writes are awaited explicitly where a phase made any (after stage_write).
  auto phase_barrier = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto lds_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

  f32x16 oa0, oa1, ob0, ob1;   // O^T accumulators: query block a/b x dh block 0/1
#pragma unroll
  for (int r = 0; r < 16; ++r) { oa0[r] = 0.f; oa1[r] = 0.f; ob0[r] = 0.f; ob1[r] = 0.f; }
  float ma = -1.0e30f, mca = -1.0e30f * scale_log2, la = 0.f, mb = -1.0e30f, mcb = -1.0e30f * scale_log2, lb = 0.f;
  f32x16 sa, sb;               // raw scores of the half in flight
  V8 pa0, pa1, pb0, pb1;       // its probabilities, packed as P.V operands
#pragma unroll
  for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
#pragma unroll
  for (int e = 0; e < 8; ++e) { pa0[e] = (T)0.f; pa1[e] = (T)0.f; pb0[e] = (T)0.f; pb1[e] = (T)0.f; }

  const int nkt = (Tn + 63) / 64;
  const int nh = (Tn + 31) / 32;          // halves holding at least one real key
  const int tr_row = 4 * hh + ((lane & 15) >> 2);
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  // ---- LDS fragments of the M phase of half h (V^T of half h - 1, K of half h): 12 reads, 32 VGPRs.  Issued at the END
  //      of the preceding V phase, in front of the barrier, so the MFMAs of the M phase start on operands that are
  //      already in registers (an s_barrier does not wait for LDS reads in flight).
  V8 vf[4], kf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { vf[i][e] = (T)0.f; kf[i][e] = (T)0.f; }
  auto frag_load = [&](int h) {
    if (!active) return;
    if (h >= 1 && h - 1 < nh) {
      const int g = h - 1;
      const T* vbase = &Vs[(g >> 1) & 1][((g & 1) * 32 + tr_row) * VLD + tr_col];
      vf[0] = lds_read_tr_pair<T>(vbase, vbase + 8 * VLD);                       // keys 0..15, dh 0..31
      vf[1] = lds_read_tr_pair<T>(vbase + 32, vbase + 8 * VLD + 32);             // keys 0..15, dh 32..63
      vf[2] = lds_read_tr_pair<T>(vbase + 16 * VLD, vbase + 24 * VLD);           // keys 16..31
      vf[3] = lds_read_tr_pair<T>(vbase + 16 * VLD + 32, vbase + 24 * VLD + 32);
    }
    if (h < nh) {
      const T* krow = &Ks[(h >> 1) & 1][((h & 1) * 32 + li) * KLD + 8 * hh];
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) kf[sl] = *reinterpret_cast<const V8*>(krow + 16 * sl);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- M phase of half h: O += V^T . P^T for half h - 1, then S = K . Q^T for half h: 16 MFMAs back to back.
  //      Keys past the end of the sequence (last half only) are masked through the accumulator's INITIAL value (-inf in
  //      their rows, 0 elsewhere: the MFMA adds it for free); full halves start from the inline constant 0.
  auto m_phase = [&](int h) {
    if (!active) return;
    __builtin_amdgcn_s_setprio(1);
    if (h >= 1 && h - 1 < nh) {
      oa0 = mfma32x32x16(vf[0], pa0, oa0);
      ob0 = mfma32x32x16(vf[0], pb0, ob0);
      oa1 = mfma32x32x16(vf[1], pa0, oa1);
      ob1 = mfma32x32x16(vf[1], pb0, ob1);
      oa0 = mfma32x32x16(vf[2], pa1, oa0);
      ob0 = mfma32x32x16(vf[2], pb1, ob0);
      oa1 = mfma32x32x16(vf[3], pa1, oa1);
      ob1 = mfma32x32x16(vf[3], pb1, ob1);
    }
    if (h < nh) {
      if (h * 32 + 32 > Tn) {
        f32x16 init;
#pragma unroll
        for (int r = 0; r < 16; ++r) init[r] = (h * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) < Tn ? 0.f : -INFINITY;
        sa = init;
        sb = init;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          sa = mfma32x32x16(kf[sl], qf0[sl], sa);
          sb = mfma32x32x16(kf[sl], qf1[sl], sb);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          sa = mfma32x32x16(kf[sl], qf0[sl], sa);
          sb = mfma32x32x16(kf[sl], qf1[sl], sb);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto v_phase = [&](int h) {            // softmax(h): scores -> packed probabilities, running max / sum
    if (!active || h < 0 || h >= nh) return;
    softmax_block<T>(sa, ma, mca, la, oa0, oa1, pa0, pa1, scale_log2, false, 0, hh, Tn);
    softmax_block<T>(sb, mb, mcb, lb, ob0, ob1, pb0, pb1, scale_log2, false, 0, hh, Tn);
  };
  // tile t + 1 goes into the other LDS buffer in phase 4t + 2 (that buffer held tile t - 1, whose last fragment reads
  // were issued before the barrier that ended phase 4t), and its global loads were issued a whole tile earlier
  auto restage = [&](int t) {
    if (t + 1 < nkt) { stage_write((t + 1) & 1); lds_done(); }
    if (t + 2 < nkt) stage_load(t + 2);
  };

  stage_load(0);
  stage_write(0);
  if (nkt > 1) stage_load(1);
  lds_done();
  __syncthreads();
  if (group_x) {
    frag_load(0);
    for (int t = 0; t < nkt; ++t) {
      m_phase(2 * t);                                      // phase 4t
      phase_barrier();
      v_phase(2 * t);                                      // phase 4t + 1
      frag_load(2 * t + 1);
      phase_barrier();
      restage(t);                                          // phase 4t + 2
      m_phase(2 * t + 1);
      phase_barrier();
      v_phase(2 * t + 1);                                  // phase 4t + 3
      frag_load(2 * t + 2);
      phase_barrier();
    }
    m_phase(2 * nkt);                                      // P.V of the last half
    phase_barrier();
    phase_barrier();
  } else {
    for (int t = 0; t < nkt; ++t) {
      v_phase(2 * t - 1);                                  // phase 4t
      frag_load(2 * t);
      phase_barrier();
      m_phase(2 * t);                                      // phase 4t + 1
      phase_barrier();
      restage(t);                                          // phase 4t + 2
      v_phase(2 * t);
      frag_load(2 * t + 1);
      phase_barrier();
      m_phase(2 * t + 1);                                  // phase 4t + 3
      phase_barrier();
    }
    v_phase(2 * nkt - 1);
    frag_load(2 * nkt);
    phase_barrier();
    m_phase(2 * nkt);
    phase_barrier();
  }

  if (!active) return;
  auto store_q = [&](int q, const f32x16& x0, const f32x16& x1, float l) {
    if (q >= Tn) return;
    const float inv = 1.0f / l;
    T* orow = out + ((long)b * Tn + q) * heads * DH + (long)head * DH;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      V4 a, c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = from_f32<T>(x0[4 * g + i] * inv);
        c[i] = from_f32<T>(x1[4 * g + i] * inv);
      }
      *reinterpret_cast<V4*>(orow + 8 * g + 4 * hh) = a;
      *reinterpret_cast<V4*>(orow + 32 + 8 * g + 4 * hh) = c;
    }
  };
  store_q(q0 + li, oa0, oa1, half_pair_sum(la));       // the two half-waves hold disjoint keys of each query
  store_q(q0 + 32 + li, ob0, ob1, half_pair_sum(lb));
}

template <class T>
static void launch_attention(const void* qkv, void* out, int B, int Tn, int heads, float scale, hipStream_t s,
                             int variant, int planar) {
  const float scale_log2 = scale * 1.4426950408889634f;
  if (variant == DSS_ATTENTION_4WAVE) {
    const int nqb = ceil_div(Tn, 256);
    hipLaunchKernelGGL((attn_fwd2_kernel<T>), dim3((unsigned)(nqb * heads * B)), dim3(256), 0, s, (const T*)qkv,
                       (T*)out, Tn, heads, B, nqb, scale_log2, planar);
  } else {
    const int nqb = ceil_div(Tn, 512);
    hipLaunchKernelGGL((attn_fwd3_kernel<T>), dim3((unsigned)(nqb * heads * B)), dim3(512), 0, s, (const T*)qkv,
                       (T*)out, Tn, heads, B, nqb, scale_log2, planar);
  }
}

}  // namespace dss

extern "C" int dss_attention_fwd(const void* qkv, int qkv_layout, void* out, int B, int T, int heads, float scale,
                                 int dtype, int variant, void* stream) {
  DSS_REQUIRE(qkv && out, "dss_attention_fwd: null pointer");
  DSS_REQUIRE(qkv_layout == DSS_ROW_MAJOR || qkv_layout == DSS_PLANAR64,
              "dss_attention_fwd: qkv_layout must be DSS_ROW_MAJOR or DSS_PLANAR64 (got %d)", qkv_layout);
  DSS_REQUIRE(B > 0 && T > 0 && heads > 0, "dss_attention_fwd: bad shape B=%d T=%d heads=%d", B, T, heads);
  DSS_REQUIRE((long)B * heads * dss::ceil_div(T, 256) < 2147483647L, "dss_attention_fwd: too many workgroups");
  DSS_REQUIRE(variant == DSS_ATTENTION_DEFAULT || variant == DSS_ATTENTION_4WAVE || variant == DSS_ATTENTION_PINGPONG,
              "dss_attention_fwd: unknown variant %d", variant);
  if (variant == DSS_ATTENTION_DEFAULT) variant = DSS_ATTENTION_PINGPONG;
  hipStream_t s = (hipStream_t)stream;
  const int planar = qkv_layout == DSS_PLANAR64;
  switch (dtype) {
    case DSS_F16: dss::launch_attention<dss::f16>(qkv, out, B, T, heads, scale, s, variant, planar); break;
    case DSS_BF16: dss::launch_attention<dss::bf16>(qkv, out, B, T, heads, scale, s, variant, planar); break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_attention_fwd: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  }
  DSS_CHECK_LAUNCH("attention");
  return DSS_OK;
}
