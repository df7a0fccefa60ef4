// attention.hip - fused multi-head self-attention for the DINO ViT blocks (head dim 64), gfx950 MFMA.
//
// Replaces DINO's Attention.forward after the qkv Linear (reached from
// extract/extract.py:94):   attn = softmax((q @ k^T) * scale) ; x = (attn @ v).transpose(1,2).reshape(B,T,C)
// The [h, T, T] score matrix is never materialised (flash-style online softmax).
//
// Two kernels:
//  1. attn_pack: qkv [B,T,3,h,64] -> Q [B,h,Tp,64], K [B,h,Tp,64], V^T [B,h,64,Tp]  (Tp = T rounded up to
//     64, zero padded).  V is transposed through LDS because the P.V contraction runs over keys, and an
//     MFMA operand wants its contraction index contiguous per lane.  Inside every 16-key group V^T's
//     keys are stored in the order the softmax registers hold them (see KEY ORDER) so one 16-byte load
//     is one MFMA operand.
//  2. attn_fwd: one wave = 32 query rows; per 32-key block
//        S^T[key][q]  = mfma_32x32x16( K-fragment , Q-fragment )      (4 MFMAs, contraction over dh=64)
//        online softmax down each lane's own query column (16 registers + one lane^32 exchange)
//        O^T[dh][q]  += mfma_32x32x16( V^T-fragment , P^T-fragment )  (4 MFMAs, contraction over 32 keys)
//     Computing the TRANSPOSED score tile makes the softmax reduction lane-local and lets the fp32
//     probabilities be packed straight into the B operand of the second MFMA - no LDS round trip.
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
// the B operand of P.V MFMA number t, i.e. operand slot (hh, e) carries key 16t + 8*(e>>2) + 4*hh + (e&3).
// V^T therefore stores, at position 16t + 8*hh + e of each 32-key block, exactly that key.
#include <stdlib.h>

#include "common.h"

namespace dss {

static constexpr int DH = 64;  // head dim of every DINO ViT

__host__ __device__ inline int attn_tp(int T) { return (T + 63) / 64 * 64; }

// position p (0..15) inside a 16-key group -> key offset inside the group
__device__ __forceinline__ int vt_key_of_pos(int p) {
  const int hh = p >> 3, e = p & 7;
  return 8 * (e >> 2) + 4 * hh + (e & 3);
}

template <class T>
__global__ __launch_bounds__(256) void attn_pack_kernel(const T* __restrict__ qkv, T* __restrict__ Qp,
                                                        T* __restrict__ Kp, T* __restrict__ Vt, int Tn,
                                                        int Tp, int heads) {
  typedef typename vec8<T>::type V8;
  __shared__ __attribute__((aligned(16))) T vtile[64][DH + 8];  // +8 halves: 16-B row skew
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * 64;
  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int row = tid >> 2;       // token inside the tile
  const int ch = (tid & 3) * 16;  // 16 halves per thread
  const int tok = t0 + row;
  const long src_row = ((long)b * Tn + tok) * 3 * heads * DH + (long)head * DH + ch;
  const long dst_row = (((long)b * heads + head) * Tp + tok) * DH + ch;
  V8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (T)0.f;
  V8 q0 = z, q1 = z, k0 = z, k1 = z, v0 = z, v1 = z;
  if (tok < Tn) {
    const T* s = qkv + src_row;
    q0 = *reinterpret_cast<const V8*>(s);
    q1 = *reinterpret_cast<const V8*>(s + 8);
    k0 = *reinterpret_cast<const V8*>(s + (long)heads * DH);
    k1 = *reinterpret_cast<const V8*>(s + (long)heads * DH + 8);
    v0 = *reinterpret_cast<const V8*>(s + 2L * heads * DH);
    v1 = *reinterpret_cast<const V8*>(s + 2L * heads * DH + 8);
  }
  *reinterpret_cast<V8*>(Qp + dst_row) = q0;
  *reinterpret_cast<V8*>(Qp + dst_row + 8) = q1;
  *reinterpret_cast<V8*>(Kp + dst_row) = k0;
  *reinterpret_cast<V8*>(Kp + dst_row + 8) = k1;
  *reinterpret_cast<V8*>(&vtile[row][ch]) = v0;
  *reinterpret_cast<V8*>(&vtile[row][ch + 8]) = v1;
  __syncthreads();
  // transposed write: thread -> (dh = tid>>2, 16 consecutive POSITIONS of the 64-key tile)
  const int dh = tid >> 2;
  const int p0 = (tid & 3) * 16;
  V8 o0, o1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    o0[i] = vtile[p0 + vt_key_of_pos(i)][dh];
    o1[i] = vtile[p0 + vt_key_of_pos(8 + i)][dh];
  }
  T* d = Vt + (((long)b * heads + head) * DH + dh) * Tp + t0 + p0;
  *reinterpret_cast<V8*>(d) = o0;
  *reinterpret_cast<V8*>(d + 8) = o1;
}

template <class T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ Qp, const T* __restrict__ Kp,
                                                       const T* __restrict__ Vt, T* __restrict__ out, int Tn,
                                                       int Tp, int heads, float scale_log2) {
  typedef typename vec8<T>::type V8;
  typedef typename vec4<T>::type V4;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31;
  const int hh = lane >> 5;
  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  if (q0 >= Tn) return;  // whole wave out of range (no block-level barrier in this kernel)
  const long bh = (long)b * heads + head;
  const T* Qb = Qp + bh * Tp * DH;
  const T* Kb = Kp + bh * Tp * DH;
  const T* Vb = Vt + bh * DH * Tp;

  V8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *reinterpret_cast<const V8*>(Qb + (long)(q0 + li) * DH + 16 * s + 8 * hh);

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -1.0e30f, lsum = 0.f;

  const int nkb = (Tn + 31) / 32;
  for (int kb = 0; kb < nkb; ++kb) {
    const int key0 = kb * 32;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const V8 kf = *reinterpret_cast<const V8*>(Kb + (long)(key0 + li) * DH + 16 * sl + 8 * hh);
      s = mfma32x32x16(kf, qf[sl], s);
    }
    const bool tail = key0 + 32 > Tn;
    float mx = -1.0e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = s[r] * scale_log2;
      if (tail) {
        const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= Tn) v = -INFINITY;
      }
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      rs += s[r];
    }
    rs += __shfl_xor(rs, 32, 64);
    lsum = lsum * alpha + rs;
    m = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    V8 pb0, pb1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pb0[e] = from_f32<T>(s[e]);
      pb1[e] = from_f32<T>(s[8 + e]);
    }
    const T* vrow0 = Vb + (long)li * Tp + key0 + 8 * hh;
    const T* vrow1 = Vb + (long)(32 + li) * Tp + key0 + 8 * hh;
    const V8 v00 = *reinterpret_cast<const V8*>(vrow0);
    const V8 v01 = *reinterpret_cast<const V8*>(vrow0 + 16);
    const V8 v10 = *reinterpret_cast<const V8*>(vrow1);
    const V8 v11 = *reinterpret_cast<const V8*>(vrow1 + 16);
    o0 = mfma32x32x16(v00, pb0, o0);
    o0 = mfma32x32x16(v01, pb1, o0);
    o1 = mfma32x32x16(v10, pb0, o1);
    o1 = mfma32x32x16(v11, pb1, o1);
  }

  const int q = q0 + li;
  if (q < Tn) {
    const float inv = 1.0f / lsum;
    T* orow = out + ((long)b * Tn + q) * heads * DH + (long)head * DH;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      V4 a, c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = from_f32<T>(o0[4 * g + i] * inv);
        c[i] = from_f32<T>(o1[4 * g + i] * inv);
      }
      *reinterpret_cast<V4*>(orow + 8 * g + 4 * hh) = a;
      *reinterpret_cast<V4*>(orow + 32 + 8 * g + 4 * hh) = c;
    }
  }
}


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
  if (tail) {
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

template <class T>
static void launch_attention(const void* qkv, void* out, int B, int Tn, int heads, float scale, void* ws,
                             hipStream_t s, int impl, int planar) {
  if (impl != 1) {  // v2 (default): LDS-staged, no pack pass, workspace unused
    const int nqb = ceil_div(Tn, 256);
    hipLaunchKernelGGL((attn_fwd2_kernel<T>), dim3((unsigned)(nqb * heads * B)), dim3(256), 0, s, (const T*)qkv,
                       (T*)out, Tn, heads, B, nqb, scale * 1.4426950408889634f, planar);
    return;
  }
  const int Tp = attn_tp(Tn);
  const size_t panel = (size_t)B * heads * Tp * DH;
  T* Qp = (T*)ws;
  T* Kp = Qp + panel;
  T* Vt = Kp + panel;
  hipLaunchKernelGGL((attn_pack_kernel<T>), dim3(Tp / 64, heads, B), dim3(256), 0, s, (const T*)qkv, Qp, Kp,
                     Vt, Tn, Tp, heads);
  const float scale_log2 = scale * 1.4426950408889634f;
  hipLaunchKernelGGL((attn_fwd_kernel<T>), dim3(ceil_div(Tn, 128), heads, B), dim3(256), 0, s, Qp, Kp, Vt,
                     (T*)out, Tn, Tp, heads, scale_log2);
}

// DSS_ATTENTION_IMPL=1 selects the v1 kernel pair (pack + register-direct); anything else the LDS-staged v2.
static int attention_impl() {
  const char* env = getenv("DSS_ATTENTION_IMPL");
  return env ? atoi(env) : 2;
}

}  // namespace dss

extern "C" size_t dss_attention_workspace_bytes(int B, int T, int heads) {
  if (B <= 0 || T <= 0 || heads <= 0) return 0;
  if (dss::attention_impl() != 1) return 0;  // the default LDS-staged kernel reads qkv in place
  return (size_t)3 * B * heads * dss::attn_tp(T) * dss::DH * 2;
}

extern "C" int dss_attention_fwd(const void* qkv, int qkv_layout, void* out, int B, int T, int heads, float scale,
                                 int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  DSS_REQUIRE(qkv && out, "dss_attention_fwd: null pointer");
  DSS_REQUIRE(qkv_layout == DSS_ROW_MAJOR || qkv_layout == DSS_PLANAR64,
              "dss_attention_fwd: qkv_layout must be DSS_ROW_MAJOR or DSS_PLANAR64 (got %d)", qkv_layout);
  DSS_REQUIRE(B > 0 && T > 0 && heads > 0, "dss_attention_fwd: bad shape B=%d T=%d heads=%d", B, T, heads);
  DSS_REQUIRE(B <= 65535 && heads <= 65535, "dss_attention_fwd: B and heads must be <= 65535");
  const size_t need = dss_attention_workspace_bytes(B, T, heads);
  if (need && (!workspace || workspace_bytes < need))
    return dss::fail(DSS_ERR_WORKSPACE, "dss_attention_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  const int impl = dss::attention_impl();
  const int planar = qkv_layout == DSS_PLANAR64;
  DSS_REQUIRE(!(planar && impl == 1), "dss_attention_fwd: DSS_ATTENTION_IMPL=1 reads interleaved qkv only");
  switch (dtype) {
    case DSS_F16: dss::launch_attention<dss::f16>(qkv, out, B, T, heads, scale, workspace, s, impl, planar); break;
    case DSS_BF16: dss::launch_attention<dss::bf16>(qkv, out, B, T, heads, scale, workspace, s, impl, planar); break;
    default: return dss::fail(DSS_ERR_BAD_ARG, "dss_attention_fwd: dtype must be DSS_F16 or DSS_BF16 (got %d)", dtype);
  }
  DSS_CHECK_LAUNCH("attention");
  return DSS_OK;
}
