// eigs_core.h - block-cooperative thick-restart Lanczos for the K smallest generalized eigenpairs of
// (D - W) v = lambda D v, solved as the K largest eigenpairs of S = D^-1/2 W D^-1/2.
//
// Replaces (reference, scipy on the host after an N^2 device->host copy):
//   extract/extract_utils.py:207-220  d = row_sum(W) ; d[d < 1e-12] = 1
//   extract/extract.py:222,227        D = diag(d) ; eigsh(D - W, k=K, sigma=0, which='LM', M=D)
//   extract/extract.py:235-240        eigenvectors.T as f32 [K, N] ; sign rule
// ARPACK (ssaupd/sseupd) is an implicitly restarted Lanczos on OP = (L - 0*D)^-1 D behind an LU of the
// dense singular L.  This file uses the mathematically equivalent restart (thick restart == implicit
// restart with exact shifts for symmetric matrices) on S directly: no LU, no dense D, no dense L, and
// each Lanczos step is one streaming pass over W.
//
// ONE WORKGROUP OWNS ONE IMAGE.  There is no inter-workgroup communication: two workgroups per CU advance 512 images
// concurrently, every block-level step is a __syncthreads().  The dominant cost is streaming W from HBM (the stored
// upper-triangular tiles, once per Lanczos step); everything else (reorthogonalisation against <= 64 basis vectors,
// the <= 64x64 Rayleigh-Ritz problem) is on-chip or L2-resident.  The convergence check of a Lanczos state does not
// stop the stream: the last wave of the workgroup solves the projected problem (WaveScope) while the other waves
// stream W for the next step; a converged image is seen one pass late and that pass is discarded.
//
// The code below is written against a tiny "block" vocabulary (DSS_TID, DSS_NT, DSS_LANE, DSS_WAVE,
// DSS_NWAVES, DSS_SYNC, DSS_WAVE_SUM, block_sum) so that the SAME source also compiles with g++ as a
// single-thread emulation (-DDSS_HOST_EMUL: one thread, one "wave" whose lane loop runs serially).
// The emulation exists only for tests/ (tests/host_emul): it lets the restart / Rayleigh-Ritz logic be
// checked against the golden vectors on a machine without a GPU.  It is never built into
// libdss_hip.so and the product never loads it.
#pragma once

#include <math.h>
#include <stdint.h>
#include <type_traits>

#ifdef DSS_HOST_EMUL
#define DSS_DEV
#define DSS_HD
#define DSS_TID 0
#define DSS_NT 1
#define DSS_LANE 0
#define DSS_WAVE 0
#define DSS_NWAVES 1
#define DSS_LANES 1
#define DSS_SYNC() ((void)0)
#define DSS_WAVE_SUM(x) (x)
#define DSS_WAVE_MAX(x) (x)
#define DSS_LDS_ADD(p, v) (*(p) += (v))
#define DSS_WAVE_SYNC() ((void)0)
#define DSS_RSQRT64(x) (1.0 / sqrt(x))
#define DSS_SETPRIO(p) ((void)0)
#define DSS_UNIFORM(p) ((void)0)
#define DSS_UNIFORM_INT(x) ((void)0)
#define DSS_F64C(x) ((double)(float)(x))
#define DSS_FRESH_F32(x) (x)
#else
#define DSS_DEV __device__ __forceinline__
#define DSS_HD __host__ __device__
// Thread / lane ids come through an opaque asm at every use: left visible, hipcc hoists every value derived from them
// (element offsets, LDS addresses, predicates of a dozen phases) out of the restart loop and keeps them live through the
// whole solver - 81 spilled VGPRs and a scratch round trip at every phase boundary (round 3); recomputing them costs a
// v_mov and an add per phase.
__device__ __forceinline__ int dss_fresh_tid() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
#define DSS_TID (dss_fresh_tid())
#define DSS_NT ((int)blockDim.x)
#define DSS_LANE (dss_fresh_tid() & 63)
#define DSS_WAVE (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))   /* wave-uniform: tile indices and their addresses stay scalar */
#define DSS_NWAVES ((int)(blockDim.x >> 6))
#define DSS_LANES 64
#define DSS_SYNC() __syncthreads()
#define DSS_WAVE_SUM(x) ::dss::wave_sum(x)
#define DSS_WAVE_MAX(x) ::dss::wave_max(x)
#define DSS_LDS_ADD(p, v) atomicAdd((p), (v))
// one wave working alone on LDS: its ds_* instructions execute in program order, so only the COMPILER has to be kept
// from moving accesses of different lanes across the point
#define DSS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define DSS_RSQRT64(x) ::dss::rsqrt64(x)
#define DSS_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
// a workgroup-uniform pointer / integer goes (back) into scalar registers here: hipcc loses the uniformity of pointers that
// are swapped inside the restart loop and then carries them - and everything derived from them - per lane
template <class T> __device__ __forceinline__ T* dss_uniform_ptr(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
#define DSS_UNIFORM(p) ((p) = dss_uniform_ptr(p))
#define DSS_UNIFORM_INT(x) ((x) = __builtin_amdgcn_readfirstlane(x))   /* a workgroup-uniform int read from LDS: keep it scalar */
// An fp64 literal that is not an inline constant occupies a register PAIR, which hipcc hoists out of every loop and keeps
// for the whole kernel (and then spills).  Thresholds and 1.5 are exact enough as an f32 literal widened where it is used.
__device__ __forceinline__ double dss_f64c(float c) { asm volatile("" : "+v"(c)); return (double)c; }
#define DSS_F64C(x) dss_f64c((float)(x))
__device__ __forceinline__ float dss_fresh_f32(float c) { asm volatile("" : "+v"(c)); return c; }
#define DSS_FRESH_F32(x) dss_fresh_f32(x)
#endif

// Lab build (-DDSS_EIGS_TIMELINE, scripts/debug/eigs_lab.py): thread 0 of every workgroup adds the shader-clock cycles
// the workgroup spends in each phase of the solve to dss_eigs_tl[phase] (slots 8..: Rayleigh-Ritz calls, Jacobi sweeps).
#if defined(DSS_EIGS_TIMELINE) && !defined(DSS_HOST_EMUL)
__device__ unsigned long long dss_eigs_tl[16];
#define DSS_ETL_DECL unsigned long long etl_t = __builtin_readcyclecounter(), etl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DSS_ETL_MARK(i) { const unsigned long long n_ = __builtin_readcyclecounter(); etl_acc[i] += n_ - etl_t; etl_t = n_; }
#define DSS_ETL_COUNT1(i) atomicAdd(&dss_eigs_tl[i], 1ull);
#define DSS_ETL_FLUSH if (DSS_TID == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&dss_eigs_tl[i_], etl_acc[i_]); }
#define DSS_ETL_WAVE_T0 const unsigned long long etl_w0 = __builtin_readcyclecounter();
#define DSS_ETL_WAVE_ADD(i) if (DSS_LANE == 0) atomicAdd(&dss_eigs_tl[i], __builtin_readcyclecounter() - etl_w0);
#else
#define DSS_ETL_DECL
#define DSS_ETL_MARK(i)
#define DSS_ETL_COUNT1(i)
#define DSS_ETL_FLUSH
#define DSS_ETL_WAVE_T0
#define DSS_ETL_WAVE_ADD(i)
#endif

#if defined(DSS_EIGS_RHO_DEBUG) && !defined(DSS_HOST_EMUL)
__device__ float* dss_eigs_rho_buf;   // [images, 64]: worst residual ratio seen by the check of state j
#define DSS_EIGS_RHO_TRACE(j, rho) if (DSS_TID == 0 && dss_eigs_rho_buf) dss_eigs_rho_buf[(size_t)blockIdx.x * 64 + (j)] = (rho);
#else
#define DSS_EIGS_RHO_TRACE(j, rho)
#endif

namespace dss {

static constexpr int EIGS_MAX_NCV = 64;
// Convergence checks: after a check whose worst residual is still > SKIP1 (SKIP2) times its bar, the next one (two)
// steps are not checked.
#ifndef DSS_EIGS_SKIP1
#define DSS_EIGS_SKIP1 30.f
#endif
#ifndef DSS_EIGS_SKIP2
#define DSS_EIGS_SKIP2 1000.f
#endif
static constexpr float EIGS_SKIP1_RATIO = DSS_EIGS_SKIP1, EIGS_SKIP2_RATIO = DSS_EIGS_SKIP2;

// Operator / selection modes (one Lanczos, three problems of the reference's _extract_eig):
//   EIGS_NORMALIZED_LAPLACIAN  S = D^-1/2 W D^-1/2, K largest theta; lambda = 1 - theta, v = D^-1/2 u
//                              (extract.py:227  eigsh(D - W, k, sigma=0, which='LM', M=D))
//   EIGS_AFFINITY_LM           operator W, K eigenpairs of largest MAGNITUDE; value = theta, v = u
//                              (extract.py:171  eigsh(W, which='LM', k); also :161-163 svd(feats) via W = F F^T)
//   EIGS_LAPLACIAN             operator W - D = -(D - W), K largest theta; lambda = -theta, v = u
//                              (extract.py:232  eigsh(D - W, k, sigma=0, which='LM'), lapnorm=False)
enum { EIGS_NORMALIZED_LAPLACIAN = 0, EIGS_AFFINITY_LM = 1, EIGS_LAPLACIAN = 2 };

struct EigsParams {
  int N;            // matrix order (patches)
  int ld;           // row stride of W in floats (multiple of 64, pad columns are zero)
  int K;            // wanted eigenpairs
  int ncv;          // Krylov dimension m, K < m <= min(N, 64)
  int keep;         // Ritz vectors kept at a restart (K <= keep <= m - 2)
  int max_restarts;
  float tol;
  int mode;         // EIGS_* above
};

// LDS carve (bytes) for one image; all offsets are multiples of 16.
struct EigsLds {
  int ldn;          // vector length padded to ld
  size_t off_xs, off_ws, off_A, off_V, off_small, total;
};
DSS_HD inline EigsLds eigs_lds_layout(int ld, int ncv) {
  EigsLds L;
  L.ldn = ld;
  size_t o = 0;
  L.off_xs = o; o += (size_t)ld * 4;
  L.off_ws = o; o += (size_t)ld * 4;
  L.off_A = o; o += (size_t)ncv * ncv * 8;
  L.off_V = o; o += (size_t)ncv * ncv * 8;
  L.off_small = o; o += 6144;
  L.total = o;
  return L;
}
// ---- packed symmetric storage of W -------------------------------------------------------------------------
// W is symmetric, so only its upper-triangular 64x64 tiles are stored and streamed, as BLOCKS of 4096 elements:
//   * full tiles: tile (I, J), I <= J < ntf, is block t = I*ntf - I*(I-1)/2 + (J - I), row-major inside the tile; diagonal
//     tiles are stored in full;
//   * the EDGE STRIP (round 5): when the last tile column holds only r = N mod 64 <= 16 columns of the matrix (N = 900: 4,
//     N = 3600 / 784 / 400: 16) it is not stored as 64-wide tiles but as MINI TILES of 64 rows x 4 columns, [row][4], mini tile
//     m = I * e4 + e holding columns 4e .. 4e+3 of the edge for tile row I = 0 .. ntf (I = ntf is the r x r corner, stored in
//     full); 4 e4 = 4 (r <= 4) or 16 columns are kept.  16 mini tiles are one block; they follow the full tiles, the last
//     block padded (its unused mini tiles are never read).  N = 900: 105 + 1 blocks = 1.07x the N (N + 1) / 2 elements of the
//     triangle, where 15 x 16 / 2 = 120 tiles were 1.21x (3.46 MB -> 1.97 MB -> 1.74 MB as float).
// Rows / columns >= N hold zeros.  ld = 64 nt stays the length of every vector of the solver.
static constexpr int WT = 64;
struct WsymLayout {
  int nt;    // tile rows / columns of the matrix: ceil(N / 64)
  int ntf;   // ... of them stored as full tiles: nt, or nt - 1 in front of an edge strip
  int e4;    // edge strip: 4 e4 columns of tile column ntf are kept (0: no strip)
};
DSS_HD inline WsymLayout wsym_layout(int N) {
  const int nt = (N + WT - 1) / WT, r = N - (N / WT) * WT;
  WsymLayout L = {nt, nt, 0};
  if (nt > 1 && r >= 1 && r <= 16) { L.ntf = nt - 1; L.e4 = r <= 4 ? 1 : 4; }
  return L;
}
DSS_HD inline int wsym_row_start(int I, int ntf) { return I * ntf - I * (I - 1) / 2; }
DSS_HD inline int wsym_full_tiles(const WsymLayout& L) { return L.ntf * (L.ntf + 1) / 2; }
DSS_HD inline int wsym_minis(const WsymLayout& L) { return L.e4 ? (L.ntf + 1) * L.e4 : 0; }
DSS_HD inline int wsym_blocks(const WsymLayout& L) { return wsym_full_tiles(L) + (wsym_minis(L) + 15) / 16; }
DSS_HD inline size_t wsym_elems(int N) { return (size_t)wsym_blocks(wsym_layout(N)) * WT * WT; }
// is column lc of tile column tj stored at all / where element (lr, lc) of tile (ti, tj), ti <= tj, lives
DSS_HD inline bool wsym_has(const WsymLayout& L, int tj, int lc) { return tj < L.ntf || lc < 4 * L.e4; }
DSS_HD inline size_t wsym_at(const WsymLayout& L, int ti, int tj, int lr, int lc) {
  if (tj < L.ntf) return ((size_t)(wsym_row_start(ti, L.ntf) + (tj - ti)) * WT + lr) * WT + lc;
  return (size_t)wsym_full_tiles(L) * WT * WT + ((size_t)(ti * L.e4 + (lc >> 2)) * WT + lr) * 4 + (lc & 3);
}

// global workspace per image (floats): two basis buffers [(ncv+1) x ld] + dis[ld]
DSS_HD inline size_t eigs_ws_floats_per_image(int ld, int ncv) { return (size_t)2 * (ncv + 1) * ld + ld; }

struct EigsSmall {  // lives in LDS at off_small (<= 6144 B)
  double alpha[EIGS_MAX_NCV];   // T diagonal
  double beta[EIGS_MAX_NCV];    // T[j][j+1] for j >= l
  double arrow[EIGS_MAX_NCV];   // T[i][l] for i < l (after a restart)
  double theta[EIGS_MAX_NCV];   // Ritz values, unsorted
  float coef[EIGS_MAX_NCV + 1]; // Gram-Schmidt coefficients
  int perm[EIGS_MAX_NCV];       // perm[rank] = column index, Ritz values descending
  float red[64];                // cross-wave reduction scratch
  double jc[EIGS_MAX_NCV / 2], js[EIGS_MAX_NCV / 2];  // Jacobi rotations of the current round
  int jp[EIGS_MAX_NCV / 2], jq[EIGS_MAX_NCV / 2];
  int flag;                     // a rotation was applied in the current Jacobi sweep
  int big;                      // ... and at least one of them was not yet negligible (see jacobi_eig)
  int nbad;                     // verdict of the convergence check that ran beside the W stream
  float rho;                    // ... and its worst residual / tolerance ratio over the K wanted pairs
};

#ifndef DSS_HOST_EMUL
// 1 / sqrt(x) for x in [1, 2] to fp64 rounding: v_rsq_f64 seed + two Newton steps (no IEEE divide / sqrt expansion)
DSS_DEV double rsqrt64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double c15 = DSS_F64C(1.5f);
  y = y * (c15 - 0.5 * x * y * y);
  y = y * (c15 - 0.5 * x * y * y);
  return y;
}
#endif

// Who executes a cooperative routine: the whole workgroup, or ONE wave on its own (the Rayleigh-Ritz check that runs
// beside the other waves' W stream, see eigs_one_image).
struct BlockScope {
  static DSS_DEV int tid() { return DSS_TID; }
  static DSS_DEV int nt() { return DSS_NT; }
  static DSS_DEV void sync() { DSS_SYNC(); }
};
struct WaveScope {
  static DSS_DEV int tid() { return DSS_LANE; }
  static DSS_DEV int nt() { return DSS_LANES; }
  static DSS_DEV void sync() { DSS_WAVE_SYNC(); }
};

DSS_DEV float block_sum(float v, EigsSmall* sm) {
#ifdef DSS_HOST_EMUL
  (void)sm;
  return v;
#else
  v = wave_sum(v);
  DSS_SYNC();
  if (DSS_LANE == 0) sm->red[DSS_WAVE] = v;
  DSS_SYNC();
  float t = 0.f;
  for (int w = 0; w < DSS_NWAVES; ++w) t += sm->red[w];
  return t;
#endif
}
DSS_DEV float block_max(float v, EigsSmall* sm) {
#ifdef DSS_HOST_EMUL
  (void)sm;
  return v;
#else
  v = wave_max(v);
  DSS_SYNC();
  if (DSS_LANE == 0) sm->red[DSS_WAVE] = v;
  DSS_SYNC();
  float t = sm->red[0];
  for (int w = 1; w < DSS_NWAVES; ++w) t = fmaxf(t, sm->red[w]);
  return t;
#endif
}

DSS_DEV float hash_unit(uint32_t e) {  // deterministic start vector entry in [-1, 1)
  uint32_t h = e * 2654435761u + 0x9E3779B9u;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

// ws = (scale ? D^-1/2 : I) * W * xs  from the packed upper-triangular tiles: the ONE pass over W per Lanczos
// step.  Every stored tile A = W[I-block, J-block] is read once and contributes A x_J to y_I and, when I != J,
// A^T x_I to y_J.  One wave owns one 16-KiB tile at a time (16 x 16-byte loads per lane, perfectly contiguous);
// the 64 row sums and 64 column sums are reduced inside the wave by halving butterflies (15 + 3 exchanges, after
// which every lane owns exactly one finished row sum and one column sum) and added to the LDS accumulator with one
// ds_add_f32 per lane.  (Floating-point LDS atomics: the accumulation order across waves is not fixed, so results
// are reproducible to rounding, not bitwise.)   xs must be zero beyond N.  Barriers inside.
// Storage type of W.  float: the values themselves.  uint16_t: round(65535 w), w in [0, 1] (normalised features,
// thresholded at zero): the normalised-Laplacian problem is invariant to the scale of W, a uniform 7.6e-6 absolute
// step perturbs the eigenvectors of the goldens by <= 1e-6 in cosine (<= 1e-7 for the K=5 cases; an f16 W would cost
// 1e-5 .. 6e-4), and the matvec - the only HBM stream of the solver - moves half the bytes.
template <class WE> struct WElem;
template <> struct WElem<float> { static constexpr float scale = 1.0f; };
template <> struct WElem<uint16_t> { static constexpr float scale = 65535.0f; };

#ifndef DSS_HOST_EMUL
// value of lane ^ M without an address register: ds_swizzle (bit-mask mode, xor inside 32 lanes) / v_permlane32_swap.
// (__shfl_xor builds a byte address per mask from the lane id; six of them live through the tile loop were six of the
// registers hipcc spilled there.)
template <int M>
DSS_DEV float lane_xor(float v, int lane) {
  if constexpr (M < 32) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (M << 10) | 0x1f));
  } else {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0]: upper half <- lower, r[1]: lower <- upper
    const unsigned lo = r[0], hi = r[1];
    return __builtin_bit_cast(float, lane < 32 ? hi : lo);
  }
}

// The reductions of one tile: rp[k] = partial of tile row 4k + g over this lane's 4 columns, c0..c3 = partials of
// columns 4q..4q+3 over this lane's 16 rows.  Halving butterflies leave every lane one finished row sum and one
// finished column sum, added to the LDS accumulator.
DSS_DEV void matvec_finish_tile(float (&rp)[16], float c0_, float c1_, float c2_, float c3_, int I, int J, int lane,
                                float* ws) {
  const int g = lane >> 4, q = lane & 15;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const bool up = (q & 8) != 0;
    const float send = up ? rp[k] : rp[k + 8];
    const float keep = up ? rp[k + 8] : rp[k];
    rp[k] = keep + lane_xor<8>(send, lane);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool up = (q & 4) != 0;
    const float send = up ? rp[k] : rp[k + 4];
    const float keep = up ? rp[k + 4] : rp[k];
    rp[k] = keep + lane_xor<4>(send, lane);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const bool up = (q & 2) != 0;
    const float send = up ? rp[k] : rp[k + 2];
    const float keep = up ? rp[k + 2] : rp[k];
    rp[k] = keep + lane_xor<2>(send, lane);
  }
  {
    const bool up = (q & 1) != 0;
    const float send = up ? rp[0] : rp[1];
    const float keep = up ? rp[1] : rp[0];
    rp[0] = keep + lane_xor<1>(send, lane);
  }
  atomicAdd(&ws[I * WT + 4 * q + g], rp[0]);   // row k = q, i.e. tile row 4 q + g
  if (I != J) {
    float c0, c1;
    {
      const bool up = (g & 1) != 0;   // lane bit 4
      const float s0 = up ? c0_ : c2_, s1 = up ? c1_ : c3_;
      const float k0 = up ? c2_ : c0_, k1 = up ? c3_ : c1_;
      c0 = k0 + lane_xor<16>(s0, lane);
      c1 = k1 + lane_xor<16>(s1, lane);
    }
    {
      const bool up = (g & 2) != 0;   // lane bit 5
      const float send = up ? c0 : c1;
      const float keep = up ? c1 : c0;
      c0 = keep + lane_xor<32>(send, lane);
    }
    atomicAdd(&ws[J * WT + 4 * q + 2 * (g & 1) + (g >> 1)], c0);   // column 4 q + 2 (g & 1) + (g >> 1)
  }
}
#endif

#ifndef DSS_HOST_EMUL
// One MINI TILE of the edge strip (64 rows x 4 columns; `raw` = this lane's row: 4 x u16 in two words, or 4 floats): tile row
// I = m / e4, edge columns 4 e .. 4 e + 3 with e = m % e4.  A lane owns one ROW: its 4 products with x_edge go to y_I[lane];
// its 4 column partials a[c] x_I[lane] are summed over the lanes by a halving butterfly (7 exchanges; column lane >> 4 ends
// in the lane) and go to y_edge.  The corner (I = ntf) is stored in full and contributes rows only.
template <class RAW>
DSS_DEV void matvec_strip_mini(const RAW& raw, int m, int e4, int ntf, const float* xs, float* ws, int lane) {
  const int I = e4 == 1 ? m : m >> 2, e = m & (e4 - 1);   // e4 is 1 or 4 (wsym_layout)
  const float xi = I < ntf ? xs[I * WT + lane] : 0.f;
  const f32x4 xj = *reinterpret_cast<const f32x4*>(xs + ntf * WT + 4 * e);
  float a0, a1, a2, a3;
  if constexpr (sizeof(RAW) == 8) {
    a0 = (float)(raw[0] & 0xffffu); a1 = (float)(raw[0] >> 16);
    a2 = (float)(raw[1] & 0xffffu); a3 = (float)(raw[1] >> 16);
  } else {
    a0 = raw[0]; a1 = raw[1]; a2 = raw[2]; a3 = raw[3];
  }
  atomicAdd(&ws[I * WT + lane], (a0 * xj[0] + a1 * xj[1]) + (a2 * xj[2] + a3 * xj[3]));
  if (I < ntf) {                                           // (wave-uniform)
    float c0 = a0 * xi, c1 = a1 * xi, c2 = a2 * xi, c3 = a3 * xi;
    {
      const bool up = (lane & 32) != 0;
      const float s0 = up ? c0 : c2, s1 = up ? c1 : c3;
      const float k0 = up ? c2 : c0, k1 = up ? c3 : c1;
      c0 = k0 + lane_xor<32>(s0, lane);
      c1 = k1 + lane_xor<32>(s1, lane);
    }
    {
      const bool up = (lane & 16) != 0;
      const float send = up ? c0 : c1;
      const float keep = up ? c1 : c0;
      c0 = keep + lane_xor<16>(send, lane);
    }
    c0 += lane_xor<8>(c0, lane);
    c0 += lane_xor<4>(c0, lane);
    c0 += lane_xor<2>(c0, lane);
    c0 += lane_xor<1>(c0, lane);
    if ((lane & 15) == 0) atomicAdd(&ws[ntf * WT + 4 * e + (lane >> 4)], c0);   // column 2 (bit 5) + (bit 4) = lane >> 4
  }
}
#endif

#ifdef DSS_EIGS_PLAIN_LOADS   // lab: W through the default cache policy instead of the streaming (nt) one
#define DSS_W_LOAD(p) (*(p))
#else
#define DSS_W_LOAD(p) __builtin_nontemporal_load(p)
#endif
struct NoSideJob { DSS_DEV void operator()() const {} };

// `side`: a job for ONE wave that does not touch xs / ws (the convergence check of the previous Lanczos state).  When
// `side_on`, the last wave of the workgroup runs it while the other waves share the tiles.
template <class WE, class Side = NoSideJob>
DSS_DEV void matvec_sym(const WE* __restrict__ Wp, int N, int ld, const float* xs, float* ws, const float* dis,
                        bool scale, bool side_on = false, const Side& side = Side()) {
  const WsymLayout LW = wsym_layout(N);            // ld == 64 LW.nt
  const int nt = LW.ntf;                           // tile rows / columns stored as full tiles
  for (int e = DSS_TID; e < ld; e += DSS_NT) ws[e] = 0.f;
  DSS_SYNC();
#ifdef DSS_HOST_EMUL
  if (side_on) side();
  for (int m = 0; m < wsym_minis(LW); ++m) {       // the edge strip: mini tile m = 64 rows x 4 columns (see wsym_layout)
    const int I = m / LW.e4, e = m % LW.e4;
    const WE* A = Wp + (size_t)wsym_full_tiles(LW) * WT * WT + (size_t)m * WT * 4;
    for (int r = 0; r < WT; ++r)
      for (int c = 0; c < 4; ++c) {
        const float a = (float)A[r * 4 + c];
        ws[I * WT + r] += a * xs[nt * WT + 4 * e + c];
        if (I < nt) ws[nt * WT + 4 * e + c] += a * xs[I * WT + r];   // (the corner I == nt is stored in full)
      }
  }
  for (int I = 0; I < nt; ++I)
    for (int J = I; J < nt; ++J) {
      const WE* A = Wp + (size_t)(wsym_row_start(I, nt) + (J - I)) * WT * WT;
      for (int r = 0; r < WT; ++r) {
        float acc = 0.f;
        for (int c = 0; c < WT; ++c) acc += (float)A[r * WT + c] * xs[J * WT + c];
        ws[I * WT + r] += acc;
      }
      if (I != J)
        for (int c = 0; c < WT; ++c) {
          float acc = 0.f;
          for (int r = 0; r < WT; ++r) acc += (float)A[r * WT + c] * xs[I * WT + r];
          ws[J * WT + c] += acc;
        }
    }
#else
  const int lane = DSS_LANE;
  const int g = lane >> 4, q = lane & 15;         // lane -> rows 4k + g (k = 0..15), columns 4q .. 4q+3
  const int ntiles = wsym_full_tiles(LW), nmini = wsym_minis(LW);
  int I = 0, row_start = 0;                        // tile row of the current tile index (advanced incrementally)
  const bool split = side_on && DSS_NWAVES > 1;
  const int stream_waves = split ? DSS_NWAVES - 1 : DSS_NWAVES;
  DSS_ETL_WAVE_T0
  if (side_on && (!split || DSS_WAVE == stream_waves)) {
    side();
    DSS_ETL_WAVE_ADD(11)   // cycles of the check (its wave)
  }
#ifndef DSS_EIGS_SIMPLE_STREAM   // lab: -DDSS_EIGS_SIMPLE_STREAM keeps the one-buffer loop below for 16-bit W too
  if constexpr (sizeof(WE) == 2) {
    {   // a fresh scalar copy of the tile base: its live range is this loop, not the whole solver (which would have it
        // spilled and reloaded per tile - a scratch reload shares vmcnt with the tile loads and drains them)
      asm volatile("" : "+s"(Wp));
    }
    // 16-bit W: TWO tiles of raw words per wave - while one is reduced, the loads of the wave's next tile are in flight.
    // The per-row-group arithmetic is one opaque asm block (4 conversions, row partial, 4 column partials): left to
    // itself hipcc converts all 64 words of a tile before the first product, and the 64 extra live registers are what
    // made every earlier double-buffered variant spill (profiles/r03_eigs_lab.txt).
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) u32x2* gptr_t;   // global: the laundered pointer must not go flat
    const gptr_t Wg = (gptr_t)(const void*)Wp;
    const auto load_tile = [&](u32x2 (&raw)[16], int t) {
      const gptr_t At = Wg + (size_t)t * (WT * WT / 4) + lane;
#pragma unroll
      for (int k = 0; k < 16; ++k) raw[k] = DSS_W_LOAD(At + k * 64);
    };
    const auto reduce_tile = [&](const u32x2 (&raw)[16], int t, u32x2 (&next)[16]) {
      while (t >= row_start + (nt - I)) { row_start += nt - I; ++I; }
      const int J = I + (t - row_start);
      int ln = lane;
      asm volatile("" : "+v"(ln));   // lane-derived addresses are RECOMPUTED per tile (a few VALU ops), not kept live
      const f32x4 xj = *reinterpret_cast<const f32x4*>(xs + J * WT + 4 * (ln & 15));
      const float* xi_base = xs + I * WT + (ln >> 4);
      float rp[16];
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float xi = xi_base[4 * k];
        float a0, a1, a2, a3;
#ifdef DSS_EIGS_ABL_NOMATH   // lab: the W stream without its arithmetic (results are garbage)
        rp[k] = __builtin_bit_cast(float, raw[k][0] ^ raw[k][1]) * xi * xj[0];
        if (false)
#endif
        asm volatile(
            "v_cvt_f32_u32_sdwa %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t"
            "v_cvt_f32_u32_sdwa %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_cvt_f32_u32_sdwa %3, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t"
            "v_cvt_f32_u32_sdwa %4, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_mul_f32 %0, %1, %11\n\t"
            "v_fmac_f32 %5, %1, %15\n\t"
            "v_fmac_f32 %0, %2, %12\n\t"
            "v_fmac_f32 %6, %2, %15\n\t"
            "v_fmac_f32 %0, %3, %13\n\t"
            "v_fmac_f32 %7, %3, %15\n\t"
            "v_fmac_f32 %0, %4, %14\n\t"
            "v_fmac_f32 %8, %4, %15"
            : "=&v"(rp[k]), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
            : "v"(raw[k][0]), "v"(raw[k][1]), "v"(xj[0]), "v"(xj[1]), "v"(xj[2]), "v"(xj[3]), "v"(xi));
        // the wave's next tile goes out once a quarter of this one is consumed: 24 + 32 raw pairs live instead of 32 + 32
        if (k == 3 && t + stream_waves < ntiles) load_tile(next, t + stream_waves);
      }
      matvec_finish_tile(rp, c0, c1, c2, c3, I, J, ln, ws);
    };
    u32x2 r0[16], r1[16];
    int t = (split && DSS_WAVE == stream_waves) ? ntiles : DSS_WAVE;
    if (t < ntiles) load_tile(r0, t);
    while (t < ntiles) {
      reduce_tile(r0, t, r1);
      t += stream_waves;
      if (t >= ntiles) break;
      reduce_tile(r1, t, r0);
      t += stream_waves;
    }
  } else
#endif
  for (int t = (split && DSS_WAVE == stream_waves) ? ntiles : DSS_WAVE; t < ntiles; t += stream_waves) {
    while (t >= row_start + (nt - I)) { row_start += nt - I; ++I; }
    const int J = I + (t - row_start);
    f32x4 a[16];
    if constexpr (sizeof(WE) == 4) {
      const f32x4* A4 = reinterpret_cast<const f32x4*>(Wp + (size_t)t * WT * WT);
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = DSS_W_LOAD(A4 + k * 64 + lane);
    } else {  // 4 x u16 per lane per row group: same (row, column) ownership as the float path, 8-byte loads
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2* A2 = reinterpret_cast<const u32x2*>(Wp + (size_t)t * WT * WT);
      u32x2 raw[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) raw[k] = DSS_W_LOAD(A2 + k * 64 + lane);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        a[k][0] = (float)(raw[k][0] & 0xffffu);
        a[k][1] = (float)(raw[k][0] >> 16);
        a[k][2] = (float)(raw[k][1] & 0xffffu);
        a[k][3] = (float)(raw[k][1] >> 16);
      }
    }
    const f32x4 xj = *reinterpret_cast<const f32x4*>(xs + J * WT + 4 * q);
    float rp[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) rp[k] = (a[k][0] * xj[0] + a[k][1] * xj[1]) + (a[k][2] * xj[2] + a[k][3] * xj[3]);
    // halving butterfly over the 16 lanes that share a row group: after step with mask w, a lane keeps the rows
    // whose k has bit (w) equal to its own lane bit
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool up = (q & 8) != 0;
      const float send = up ? rp[k] : rp[k + 8];
      const float keep = up ? rp[k + 8] : rp[k];
      rp[k] = keep + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool up = (q & 4) != 0;
      const float send = up ? rp[k] : rp[k + 4];
      const float keep = up ? rp[k + 4] : rp[k];
      rp[k] = keep + __shfl_xor(send, 4, 64);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool up = (q & 2) != 0;
      const float send = up ? rp[k] : rp[k + 2];
      const float keep = up ? rp[k + 2] : rp[k];
      rp[k] = keep + __shfl_xor(send, 2, 64);
    }
    {
      const bool up = (q & 1) != 0;
      const float send = up ? rp[0] : rp[1];
      const float keep = up ? rp[1] : rp[0];
      rp[0] = keep + __shfl_xor(send, 1, 64);
    }
    // this lane now owns row k = q (bit-for-bit: bit 3 of k from lane bit 3, ...), i.e. tile row 4*q + g
    atomicAdd(&ws[I * WT + 4 * q + g], rp[0]);
    if (I != J) {
      f32x4 cp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float xi = xs[I * WT + 4 * k + g];
        cp[0] += a[k][0] * xi; cp[1] += a[k][1] * xi; cp[2] += a[k][2] * xi; cp[3] += a[k][3] * xi;
      }
      // reduce over the 4 row groups (lane bits 4,5) while halving 4 -> 2 -> 1 columns per lane
      float c0, c1;
      {
        const bool up = (g & 1) != 0;   // lane bit 4
        const float s0 = up ? cp[0] : cp[2], s1 = up ? cp[1] : cp[3];
        const float k0 = up ? cp[2] : cp[0], k1 = up ? cp[3] : cp[1];
        c0 = k0 + __shfl_xor(s0, 16, 64);
        c1 = k1 + __shfl_xor(s1, 16, 64);
      }
      {
        const bool up = (g & 2) != 0;   // lane bit 5
        const float send = up ? c0 : c1;
        const float keep = up ? c1 : c0;
        c0 = keep + __shfl_xor(send, 32, 64);
      }
      // owned column: 4*q + 2*(g & 1) + (g >> 1)
      atomicAdd(&ws[J * WT + 4 * q + 2 * (g & 1) + (g >> 1)], c0);
    }
  }
  // The edge strip, behind the tiles: its mini tiles (512 B of 16-bit W each) go round-robin to the streaming waves from the
  // LAST one down - tile t went to wave t mod stream_waves, so those are the waves with one tile less (N = 900: 105 tiles over
  // 8 waves = 14 for wave 0, 13 for the others, and the 15 mini tiles are 2-3 per wave of a fifth of a tile's work each).
  if (LW.e4) {
    typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
    typedef typename std::conditional<sizeof(WE) == 4, f32x4, u32x2s>::type raw_t;
    typedef const __attribute__((address_space(1))) raw_t* gsp_t;
    int ln = lane;
    asm volatile("" : "+v"(ln));                           // lane-derived addresses are computed HERE, not kept live over the tile loop
    const gsp_t S = (gsp_t)(const void*)(Wp + (size_t)ntiles * WT * WT) + ln;
    int m = stream_waves - 1 - DSS_WAVE;                   // (negative for the wave of the side job)
    if (m >= 0 && m < nmini) {
      raw_t cur = DSS_W_LOAD(S + m * 64);
      for (;;) {
        const int mn = m + stream_waves;
        raw_t nxt = cur;
        if (mn < nmini) nxt = DSS_W_LOAD(S + mn * 64);     // the next one is in flight under this one's butterflies
        matvec_strip_mini(cur, m, LW.e4, nt, xs, ws, ln);
        if (mn >= nmini) break;
        cur = nxt;
        m = mn;
      }
    }
  }
  if (DSS_WAVE == 0) { DSS_ETL_WAVE_ADD(side_on ? 12 : 10) }   // tile loop of wave 0: with / without a check beside it
#endif
  DSS_SYNC();
  if (DSS_WAVE == 0) { DSS_ETL_WAVE_ADD(side_on ? 14 : 13) }     // ... including the wait for the slowest wave
  if (scale)
    for (int e = DSS_TID; e < N; e += DSS_NT) ws[e] *= dis[e];
  DSS_SYNC();
}

#ifndef DSS_HOST_EMUL
// Sum R values (a power of two <= 32) over the 64 lanes with R + 4 exchanges instead of 6 R: a halving butterfly (a lane
// keeps half of its values and adds the partner's other half) leaves value k complete in the lanes k * (64 / R) ...,
// which a last plain butterfly over the low lane bits makes identical.  Returns the value of index lane / (64 / R).
template <int R>
DSS_DEV float wave_sum_many(float (&p)[R], int lane) {
  int mask = 32;
#pragma unroll
  for (int n = R / 2; n >= 1; n >>= 1, mask >>= 1) {
    const bool up = (lane & mask) != 0;
#pragma unroll
    for (int k = 0; k < n; ++k) {
      const float send = up ? p[k] : p[k + n];
      const float keep = up ? p[k + n] : p[k];
      p[k] = keep + __shfl_xor(send, mask, 64);
    }
  }
#pragma unroll
  for (; mask >= 1; mask >>= 1) p[0] += __shfl_xor(p[0], mask, 64);
  return p[0];
}
#endif

// Gram-Schmidt against the L2-resident basis.  Threads own ELEMENTS (e, e + NT), and a round handles 8 basis vectors:
// 16 independent coalesced loads per thread are in flight before the first use (a loop over the elements of one
// vector, a wave per vector, costs one L2 round trip per 64 elements: 14 in a row at N = 900).
//
// coef[i] = V[i] . ws  for i in [0, nvec).  Barriers inside; coef is complete on return.
DSS_DEV void basis_dots(const float* V, int ldv, int nvec, const float* ws, int N, float* coef) {
  for (int i = DSS_TID; i < nvec; i += DSS_NT) coef[i] = 0.f;
  DSS_SYNC();
  for (int i0 = 0; i0 < nvec; i0 += 8) {
    float part[8];
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) part[ii] = 0.f;
    for (int e = DSS_TID; e < N; e += 2 * DSS_NT) {
      const bool two = e + DSS_NT < N;
      const int e1 = two ? e + DSS_NT : e;
      const float w0 = ws[e], w1 = two ? ws[e1] : 0.f;
      float v0[8], v1[8];
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) {
        const float* v = V + (size_t)(i0 + ii < nvec ? i0 + ii : nvec - 1) * ldv;   // clamped: no divergent loads
        v0[ii] = v[e];
        v1[ii] = v[e1];
      }
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) part[ii] += v0[ii] * w0 + v1[ii] * w1;
    }
#ifdef DSS_HOST_EMUL
    for (int ii = 0; ii < 8; ++ii)
      if (i0 + ii < nvec) coef[i0 + ii] += part[ii];
#else
    const float t = wave_sum_many<8>(part, DSS_LANE);
    if ((DSS_LANE & 7) == 0 && i0 + (DSS_LANE >> 3) < nvec) DSS_LDS_ADD(&coef[i0 + (DSS_LANE >> 3)], t);
#endif
  }
  DSS_SYNC();
}

// ws[e] -= sum_i coef[i] * V[i][e]; returns this thread's share of |ws|^2 after the update
DSS_DEV float basis_axpy(const float* V, int ldv, int nvec, float* ws, int N, const float* coef) {
  float n2 = 0.f;
  for (int e = DSS_TID; e < N; e += 2 * DSS_NT) {
    const bool two = e + DSS_NT < N;
    const int e1 = two ? e + DSS_NT : e;
    float a0 = ws[e], a1 = ws[e1];
    for (int i0 = 0; i0 < nvec; i0 += 8) {
      float v0[8], v1[8];
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) {
        const float* v = V + (size_t)(i0 + ii < nvec ? i0 + ii : nvec - 1) * ldv;
        v0[ii] = v[e];
        v1[ii] = v[e1];
      }
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) {
        const float c = i0 + ii < nvec ? coef[i0 + ii] : 0.f;
        a0 -= c * v0[ii];
        a1 -= c * v1[ii];
      }
    }
    ws[e] = a0;
    n2 += a0 * a0;
    if (two) { ws[e1] = a1; n2 += a1 * a1; }
  }
  return n2;
}

// Symmetric eigen-decomposition of the (m x m) projected matrix by the classical two-sided Jacobi method with
// a parallel (round-robin) ordering, fp64, entirely in LDS.  A round rotates M/2 DISJOINT index pairs at once:
//   1. one thread per pair: (c, s) from a_pp, a_qq, a_pq                       - no reductions at all
//   2. one thread per (pair, pair): a 2x2 block of A <- J^T A J and two rows of two columns of V <- V J
// with a barrier after each step.  A and Vr are column-major with leading dimension m (A is symmetric).
// On exit theta[c] = A[c][c], Vr[:, c] = eigenvector, perm = ranks (descending theta).
template <class S>
DSS_DEV void jacobi_eig(double* A, double* Vr, int m, EigsSmall* sm, bool by_magnitude) {
  const int M = (m + 1) & ~1, np = M / 2;
  const float inv_np = 1.0f / (float)np;
  for (int sweep = 0; sweep < 30; ++sweep) {
    if (S::tid() == 0) { sm->flag = 0; sm->big = 0; }
    S::sync();
    for (int round = 0; round < M - 1; ++round) {
      // 1. rotation of every pair from three entries.  Only ORTHOGONALITY needs fp64 (c^2 + s^2 = 1): an angle that is
      //    off by 1e-7 relative still shrinks a_pq by that factor, so tan comes from f32 arithmetic and the long fp64
      //    divide / square-root chains (the latency of a round, with <= 32 lanes busy) reduce to one reciprocal root.
      for (int k = S::tid(); k < np; k += S::nt()) {
        int p, q;
        if (k == 0) { p = M - 1; q = round; }
        else {
          p = round + k; if (p >= M - 1) p -= M - 1;
          q = round - k; if (q < 0) q += M - 1;
        }
        if (p > q) { const int t = p; p = q; q = t; }
        double c = 1.0, sn = 0.0;
        if (q < m) {   // q == m: the dummy index of an odd m - identity
          const double app = A[(size_t)p * m + p], aqq = A[(size_t)q * m + q], apq = A[(size_t)q * m + p];
          const double a2 = apq * apq, d2 = fabs(app * aqq);
          const double tiny = DSS_F64C(1e-34f);
          if (a2 > tiny + DSS_F64C(1e-20f) * d2) sm->big = 1;  // |a_pq| > 1e-10 sqrt|a_pp a_qq|; benign race: every writer stores 1
          if (a2 > tiny + DSS_F64C(1e-30f) * d2) {             // |a_pq| > 1e-15 sqrt|a_pp a_qq|
            const float h = (float)(0.5 * (aqq - app)), bq = (float)apq;
            const float tf = bq / (fabsf(h) + sqrtf(h * h + bq * bq));   // tan of the smaller rotation angle
            const double t = (double)(h < 0.f ? -tf : tf);
            c = DSS_RSQRT64(1.0 + t * t);
            sn = c * t;
            sm->flag = 1;  // benign race: every writer stores 1
          }
        }
        sm->jp[k] = p; sm->jq[k] = q; sm->jc[k] = c; sm->js[k] = sn;
      }
      S::sync();
      // 2. A <- J^T A J by 2x2 blocks (row pair k1 x column pair k2: four entries in, four out, nobody else touches
      //    them) and V <- V J (row pair r = 2 k1, 2 k1 + 1 x column pair k2) in the same pass: one barrier per round.
      for (int idx = S::tid(); idx < np * np; idx += S::nt()) {
        const int k1 = (int)(((float)idx + 0.5f) * inv_np), k2 = idx - k1 * np;
        const int p2 = sm->jp[k2], q2 = sm->jq[k2];
        const double c2 = sm->jc[k2], s2 = sm->js[k2];
        const bool hq2 = q2 < m;
        {
          const int p1 = sm->jp[k1], q1 = sm->jq[k1];
          const double c1 = sm->jc[k1], s1 = sm->js[k1];
          const bool hq1 = q1 < m;
          const double app = A[(size_t)p2 * m + p1];
          const double apq = hq2 ? A[(size_t)q2 * m + p1] : 0.;
          const double aqp = hq1 ? A[(size_t)p2 * m + q1] : 0.;
          const double aqq = hq1 && hq2 ? A[(size_t)q2 * m + q1] : 0.;
          // columns (J on the right), then rows (J^T on the left)
          const double bpp = c2 * app - s2 * apq, bpq = s2 * app + c2 * apq;
          const double bqp = c2 * aqp - s2 * aqq, bqq = s2 * aqp + c2 * aqq;
          A[(size_t)p2 * m + p1] = c1 * bpp - s1 * bqp;
          if (hq2) A[(size_t)q2 * m + p1] = c1 * bpq - s1 * bqq;
          if (hq1) A[(size_t)p2 * m + q1] = s1 * bpp + c1 * bqp;
          if (hq1 && hq2) A[(size_t)q2 * m + q1] = s1 * bpq + c1 * bqq;
        }
        if (hq2) {
          const int r0 = 2 * k1, r1 = 2 * k1 + 1;
          const double v0p = Vr[(size_t)p2 * m + r0], v0q = Vr[(size_t)q2 * m + r0];
          Vr[(size_t)p2 * m + r0] = c2 * v0p - s2 * v0q;
          Vr[(size_t)q2 * m + r0] = s2 * v0p + c2 * v0q;
          if (r1 < m) {
            const double v1p = Vr[(size_t)p2 * m + r1], v1q = Vr[(size_t)q2 * m + r1];
            Vr[(size_t)p2 * m + r1] = c2 * v1p - s2 * v1q;
            Vr[(size_t)q2 * m + r1] = s2 * v1p + c2 * v1q;
          }
        }
      }
      S::sync();
    }
    if (S::tid() == 0) { DSS_ETL_COUNT1(9) }
    // Converged after a sweep without a rotation - or one whose rotations all had |a_pq| <= 1e-10 sqrt|a_pp a_qq|: the
    // off-diagonal left behind is second order in that (<= 1e-20 / gap), so the confirming sweep would not rotate
    // anything that matters at the 2e-6 residual test.  (uniform: read after the barrier)
    if (sm->flag == 0 || sm->big == 0) break;
    S::sync();
  }
  if (S::tid() == 0) { DSS_ETL_COUNT1(8) }
  for (int c = S::tid(); c < m; c += S::nt()) sm->theta[c] = A[(size_t)c * m + c];
  S::sync();
  for (int c = S::tid(); c < m; c += S::nt()) {
    int rank = 0;
    const double tc = by_magnitude ? fabs(sm->theta[c]) : sm->theta[c];
    for (int j = 0; j < m; ++j) {
      const double tj = by_magnitude ? fabs(sm->theta[j]) : sm->theta[j];
      rank += (tj > tc || (tj == tc && j < c)) ? 1 : 0;
    }
    sm->perm[rank] = c;
  }
  S::sync();
}

// Rayleigh-Ritz on the projected matrix T (m x m): diagonal alpha, arrow column l (after a restart),
// off-diagonal beta[j] = T[j][j+1] for j >= l.  Returns the number of the K wanted Ritz pairs whose residual
// |beta_last * z_{m-1,i}| exceeds tol * max(|theta_i|, 1e-3).  Leaves Vr / sm->theta / sm->perm set.
template <class S>
DSS_DEV int rayleigh_ritz(double* A, double* Vr, int m, int l, int K, double beta_last, float tol, EigsSmall* sm,
                          bool by_magnitude, float* worst_ratio = nullptr) {
  for (int idx = S::tid(); idx < m * m; idx += S::nt()) {
    const int c = idx / m, r = idx - c * m;
    double t = 0.;
    if (r == c) t = sm->alpha[r];
    else {
      const int lo = r < c ? r : c, hi = r < c ? c : r;
      if (hi == l && lo < l) t = sm->arrow[lo];
      else if (hi == lo + 1 && lo >= l) t = sm->beta[lo];
    }
    A[idx] = t;
    Vr[idx] = r == c ? 1.0 : 0.0;
  }
  S::sync();
  jacobi_eig<S>(A, Vr, m, sm, by_magnitude);
  double tmax = 0.;  // tolerance floor relative to the largest Ritz value in magnitude (1 for the normalised Laplacian)
  for (int c = 0; c < m; ++c) tmax = fabs(sm->theta[c]) > tmax ? fabs(sm->theta[c]) : tmax;
  int nbad = 0;
  double worst = 0.;
  for (int i = 0; i < K; ++i) {
    if (i >= m) { ++nbad; worst = DSS_F64C(1e30f); continue; }
    const int c = sm->perm[i];
    const double res = fabs(beta_last * Vr[(size_t)c * m + (m - 1)]);
    const double th = fabs(sm->theta[c]);
    const double floor_ = DSS_F64C(1e-3f) * tmax;
    const double bar = (double)DSS_FRESH_F32(tol) * (th > floor_ ? th : floor_);
    if (res > bar) ++nbad;
    if (res > worst * bar) worst = res / bar;
  }
  if (worst_ratio) *worst_ratio = (float)(worst < DSS_F64C(1e30f) ? worst : DSS_F64C(1e30f));
  return nbad;
}

// The whole eigen stage for ONE image (called by every thread of the owning workgroup).
//   W          packed upper-triangular storage of the symmetric non-negative affinity (wsym_elems(N) elements, see wsym_layout)
//   gws        global workspace of eigs_ws_floats_per_image(ld, ncv) floats
//   lds        LDS block of eigs_lds_layout(ld, ncv).total bytes (16-byte aligned)
//   eigenvalues[K], eigenvectors[K, N] outputs; *info = +passes (converged) / -passes (budget exhausted)
template <class WE>
DSS_DEV void eigs_one_image(const WE* __restrict__ W, const EigsParams P, float* gws, unsigned char* lds,
                            float* eigenvalues, float* eigenvectors, int32_t* info) {
  const int N = P.N, ld = P.ld, K = P.K, mmax = P.ncv;
  const EigsLds L = eigs_lds_layout(ld, mmax);
  float* xs = reinterpret_cast<float*>(lds + L.off_xs);
  float* ws = reinterpret_cast<float*>(lds + L.off_ws);
  double* A = reinterpret_cast<double*>(lds + L.off_A);
  double* Vr = reinterpret_cast<double*>(lds + L.off_V);
  EigsSmall* sm = reinterpret_cast<EigsSmall*>(lds + L.off_small);
  const int ldv = ld;
  float* Va = gws;                                   // the basis in use; the other of the two buffers (a restart's target) is
                                                     // derived where it is needed, not carried through the solver
  float* dis = gws + (size_t)2 * (mmax + 1) * ldv;
  int passes = 0;
  DSS_ETL_DECL

  // ---- degree: d = W 1 ; clamp (extract_utils.py:218) ; dis = d^-1/2 (normalised) or d itself (plain Laplacian) ----
  const int mode = P.mode;
  const bool by_mag = mode == EIGS_AFFINITY_LM;
  if (mode != EIGS_AFFINITY_LM) {
    for (int e = DSS_TID; e < ld; e += DSS_NT) xs[e] = e < N ? 1.0f : 0.0f;
    DSS_SYNC();
    matvec_sym(W, N, ld, xs, ws, nullptr, false);
    ++passes;
    for (int e = DSS_TID; e < ld; e += DSS_NT) {
      float d = e < N ? ws[e] : 1.0f;
      if (d < 1e-12f) d = 1.0f;
      dis[e] = e >= N ? 0.0f : (mode == EIGS_NORMALIZED_LAPLACIAN ? 1.0f / sqrtf(d) : d);
    }
  }
  // ---- start vector -------------------------------------------------------------------------------
  float nrm2 = 0.f;
  for (int e = DSS_TID; e < N; e += DSS_NT) { const float v = hash_unit((uint32_t)e); ws[e] = v; nrm2 += v * v; }
  nrm2 = block_sum(nrm2, sm);
  {
    const float inv = 1.0f / sqrtf(nrm2);
    for (int e = DSS_TID; e < N; e += DSS_NT) Va[e] = ws[e] * inv;
  }
  DSS_SYNC();  // Va[0], dis visible to the block (same-workgroup global writes + barrier)
  DSS_ETL_MARK(0)

  int l = 0;          // kept Ritz vectors (0 on the first cycle)
  int m = mmax;       // effective Krylov dimension of this cycle (shrinks on breakdown)
  double beta_last = 0.;
  bool converged = false;
  int restart = 0;
  for (;; ++restart) {
    // ---- extend the Krylov basis from l to m --------------------------------------------------------
    bool breakdown = false, early = false;
    int next_check = 0;
    for (int j = l; j < mmax; ++j) {
      DSS_UNIFORM(Va); DSS_UNIFORM(dis);
      const float* vj = Va + (size_t)j * ldv;
      // Convergence check of the CURRENT state (j basis vectors, T_j, beta_last = beta_{j-1}) by the last wave, beside
      // the W stream of step j (the Jacobi sweeps are latency-bound and need no bandwidth).  A converged image is
      // detected one pass late - that pass is discarded - instead of stalling every step of every image.
      const bool check = j >= K + 3 && j > l + 1 && j >= next_check;
      const auto side = [&]() {
        float rho;
        DSS_SETPRIO(3);   // a chain of dependent instructions sharing its SIMD with three streaming waves: issue it first
        const int nb = rayleigh_ritz<WaveScope>(A, Vr, j, l, K, beta_last, P.tol, sm, by_mag, &rho);
        DSS_SETPRIO(0);
        if (DSS_LANE == 0) { sm->nbad = nb; sm->rho = rho; }
      };
      {
        const bool nl = mode == EIGS_NORMALIZED_LAPLACIAN;
        for (int e = DSS_TID; e < ld; e += DSS_NT) xs[e] = e < N ? (nl ? dis[e] * vj[e] : vj[e]) : 0.0f;
        DSS_SYNC();
        matvec_sym(W, N, ld, xs, ws, dis, nl, check, side);     // w = D^-1/2 W D^-1/2 v   or   w = W v
      }
      ++passes;
      if (check) {   // residuals of a Lanczos process fall by a bounded factor per step: far from the bar, skip checks
        const float rho = sm->rho;
        next_check = j + 1 + (rho > EIGS_SKIP2_RATIO ? 2 : (rho > EIGS_SKIP1_RATIO ? 1 : 0));
        DSS_UNIFORM_INT(next_check);
        DSS_EIGS_RHO_TRACE(j, rho)
      }
      if (check && sm->nbad == 0) {   // Vr / theta / perm describe T_j: finish from the j-vector basis
        m = j;
        early = true;
        DSS_ETL_MARK(1)
        break;
      }
      if (mode == EIGS_LAPLACIAN) {                             // w = W v - D v = -(D - W) v
        for (int e = DSS_TID; e < N; e += DSS_NT) ws[e] -= dis[e] * vj[e];
        DSS_SYNC();
      }
      DSS_ETL_MARK(1)
      // classical Gram-Schmidt, two passes (full reorthogonalisation against V[0..j])
      basis_dots(Va, ldv, j + 1, ws, N, sm->coef);
      double alpha = (double)sm->coef[j];
      basis_axpy(Va, ldv, j + 1, ws, N, sm->coef);
      DSS_SYNC();
      basis_dots(Va, ldv, j + 1, ws, N, sm->coef);
      alpha += (double)sm->coef[j];
      float b2 = basis_axpy(Va, ldv, j + 1, ws, N, sm->coef);   // every thread re-reads only its own elements below
      b2 = block_sum(b2, sm);
      const float beta = sqrtf(b2);
      if (DSS_TID == 0) { sm->alpha[j] = alpha; sm->beta[j] = (double)beta; }
      beta_last = (double)beta;
      if (beta < 1e-5f || j + 1 >= N) {  // invariant subspace (or the whole space) reached
        m = j + 1;
        breakdown = true;
        beta_last = 0.;
        DSS_SYNC();
        break;
      }
      {
        const float inv = 1.0f / beta;
        float* vn = Va + (size_t)(j + 1) * ldv;
        for (int e = DSS_TID; e < N; e += DSS_NT) vn[e] = ws[e] * inv;
      }
      m = j + 1;
      DSS_SYNC();
      DSS_ETL_MARK(2)
    }
    // ---- Rayleigh-Ritz on the full basis (skipped when a mid-cycle check already converged) -----------------
    int nbad = 0;
    if (!early) nbad = rayleigh_ritz<BlockScope>(A, Vr, m, l, K, beta_last, P.tol, sm, by_mag);
    DSS_ETL_MARK(4)
    converged = (nbad == 0);
    if (converged || breakdown || restart >= P.max_restarts) break;
    // ---- thick restart: keep the best `keep` Ritz vectors -----------------------------------------------
    float* Vb = Va == gws ? gws + (size_t)(mmax + 1) * ldv : gws;
    DSS_UNIFORM(Va); DSS_UNIFORM(Vb);
    int keep = P.keep < m - 2 ? P.keep : m - 2;
    if (keep < K) keep = K < m - 1 ? K : m - 1;
    DSS_SYNC();
    float* Zf = reinterpret_cast<float*>(A);  // A is dead after the Ritz values were taken: reuse as f32 Z
    for (int idx = DSS_TID; idx < m * keep; idx += DSS_NT) {
      const int i = idx / m, j = idx - i * m;  // Zf[i][j] = Z[j][perm[i]]
      Zf[idx] = (float)Vr[(size_t)sm->perm[i] * m + j];
    }
    DSS_SYNC();
    for (int e = DSS_TID; e < N; e += DSS_NT) {
      for (int i0 = 0; i0 < keep; i0 += 8) {
        float acc[8];
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) acc[ii] = 0.f;
        for (int j = 0; j < m; ++j) {
          const float vj = Va[(size_t)j * ldv + e];
#pragma unroll
          for (int ii = 0; ii < 8; ++ii)
            if (i0 + ii < keep) acc[ii] += vj * Zf[(size_t)(i0 + ii) * m + j];
        }
#pragma unroll
        for (int ii = 0; ii < 8; ++ii)
          if (i0 + ii < keep) Vb[(size_t)(i0 + ii) * ldv + e] = acc[ii];
      }
      Vb[(size_t)keep * ldv + e] = Va[(size_t)m * ldv + e];
    }
    DSS_SYNC();
    for (int i = DSS_TID; i < keep; i += DSS_NT) {  // alpha/arrow do not alias theta/Vr
      const int c = sm->perm[i];
      sm->alpha[i] = sm->theta[c];
      sm->arrow[i] = beta_last * Vr[(size_t)c * m + (m - 1)];
    }
    Va = Vb;
    DSS_UNIFORM(Va);
    l = keep;
    DSS_SYNC();
    DSS_ETL_MARK(5)
  }

  // ---- Ritz vectors -> generalized eigenvectors v = D^-1/2 u, sign rule, eigenvalues ----------------------
  DSS_SYNC();
  DSS_UNIFORM(Va); DSS_UNIFORM(dis); DSS_UNIFORM(eigenvectors);
  const float vscale = sqrtf(WElem<WE>::scale);
  float* Zf = reinterpret_cast<float*>(A);
  for (int idx = DSS_TID; idx < m * K; idx += DSS_NT) {
    const int i = idx / m, j = idx - i * m;
    Zf[idx] = (float)Vr[(size_t)sm->perm[i] * m + j];
  }
  DSS_SYNC();
  for (int i = 0; i < K; ++i) {
    int pos = 0;
    for (int e = DSS_TID; e < N; e += DSS_NT) {
      float u = 0.f;
      for (int j = 0; j < m; ++j) u += Va[(size_t)j * ldv + e] * Zf[(size_t)i * m + j];
      // D was accumulated in storage units (scale * true degree): v^T D_true v = 1 needs the sqrt(scale) back
      const float v = mode == EIGS_NORMALIZED_LAPLACIAN ? u * dis[e] * vscale : u;
      ws[e] = v;
      pos += v > 0.f ? 1 : 0;
    }
    const int cnt = (int)(block_sum((float)pos, sm) + 0.5f);
    // extract/extract.py:238-240: negate iff 0.5 < mean(v > 0) < 1.0  <=>  2*cnt > N and cnt < N
    const float sgn = (2 * cnt > N && cnt < N) ? -1.0f : 1.0f;
    for (int e = DSS_TID; e < N; e += DSS_NT) eigenvectors[(size_t)i * N + e] = sgn * ws[e];
    DSS_SYNC();
  }
  for (int i = DSS_TID; i < K; i += DSS_NT) {
    const double th = sm->theta[sm->perm[i]];
    eigenvalues[i] = (float)(mode == EIGS_NORMALIZED_LAPLACIAN ? 1.0 - th : (mode == EIGS_LAPLACIAN ? -th : th));
  }
  if (DSS_TID == 0) *info = converged ? passes : -passes;
  DSS_ETL_MARK(6)
  DSS_ETL_FLUSH
}

}  // namespace dss
