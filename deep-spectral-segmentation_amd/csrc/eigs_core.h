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
// ONE WORKGROUP OWNS ONE IMAGE.  There is no inter-workgroup communication: 256 CUs advance 256 images
// concurrently, every block-level step is a __syncthreads().  The dominant cost is streaming W from
// HBM (4*N*ld bytes per Lanczos step; HBM-bound); everything else (reorthogonalisation against <= 64
// basis vectors, the <= 64x64 Rayleigh-Ritz problem) is on-chip or L2-resident.
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
#else
#define DSS_DEV __device__ __forceinline__
#define DSS_HD __host__ __device__
#define DSS_TID ((int)threadIdx.x)
#define DSS_NT ((int)blockDim.x)
#define DSS_LANE ((int)(threadIdx.x & 63))
#define DSS_WAVE ((int)(threadIdx.x >> 6))
#define DSS_NWAVES ((int)(blockDim.x >> 6))
#define DSS_LANES 64
#define DSS_SYNC() __syncthreads()
#define DSS_WAVE_SUM(x) ::dss::wave_sum(x)
#define DSS_WAVE_MAX(x) ::dss::wave_max(x)
#endif

namespace dss {

static constexpr int EIGS_MAX_NCV = 64;

struct EigsParams {
  int N;            // matrix order (patches)
  int ld;           // row stride of W in floats (multiple of 64, pad columns are zero)
  int K;            // wanted eigenpairs
  int ncv;          // Krylov dimension m, K < m <= min(N, 64)
  int keep;         // Ritz vectors kept at a restart (K <= keep <= m - 2)
  int max_restarts;
  float tol;
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
  int flag;
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

// ws[r] = (scale ? dis[r] : 1) * sum_c W[r][c] * xs[c]     - the one pass over W per Lanczos step.
DSS_DEV void matvec_rows(const float* __restrict__ W, int N, int ld, const float* xs, float* ws,
                         const float* dis, bool scale) {
#ifdef DSS_HOST_EMUL
  for (int r = 0; r < N; ++r) {
    float acc = 0.f;
    for (int c = 0; c < ld; ++c) acc += W[(size_t)r * ld + c] * xs[c];
    ws[r] = scale ? acc * dis[r] : acc;
  }
#else
  // One wave owns 4 consecutive rows at a time: 4 x 16-byte global loads per lane share one 16-byte LDS
  // read of x; 64 lanes x float4 = 1 KiB of each row per instruction (fully coalesced).
  const int lane = DSS_LANE;
  const int nvec = ld >> 2;
  const f32x4* xs4 = reinterpret_cast<const f32x4*>(xs);
  for (int r0 = DSS_WAVE * 4; r0 < N; r0 += DSS_NWAVES * 4) {
    const int r1 = r0 + 1 < N ? r0 + 1 : N - 1, r2 = r0 + 2 < N ? r0 + 2 : N - 1,
              r3 = r0 + 3 < N ? r0 + 3 : N - 1;
    const f32x4* w0 = reinterpret_cast<const f32x4*>(W + (size_t)r0 * ld);
    const f32x4* w1 = reinterpret_cast<const f32x4*>(W + (size_t)r1 * ld);
    const f32x4* w2 = reinterpret_cast<const f32x4*>(W + (size_t)r2 * ld);
    const f32x4* w3 = reinterpret_cast<const f32x4*>(W + (size_t)r3 * ld);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
    for (int c = lane; c < nvec; c += 64) {
      const f32x4 x = xs4[c];
      const f32x4 p0 = __builtin_nontemporal_load(w0 + c);
      const f32x4 p1 = __builtin_nontemporal_load(w1 + c);
      const f32x4 p2 = __builtin_nontemporal_load(w2 + c);
      const f32x4 p3 = __builtin_nontemporal_load(w3 + c);
      a0 += (p0[0] * x[0] + p0[1] * x[1]) + (p0[2] * x[2] + p0[3] * x[3]);
      a1 += (p1[0] * x[0] + p1[1] * x[1]) + (p1[2] * x[2] + p1[3] * x[3]);
      a2 += (p2[0] * x[0] + p2[1] * x[1]) + (p2[2] * x[2] + p2[3] * x[3]);
      a3 += (p3[0] * x[0] + p3[1] * x[1]) + (p3[2] * x[2] + p3[3] * x[3]);
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    if (lane < 4) {
      const int r = r0 + lane;
      if (r < N) {
        float a = lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : a3));
        ws[r] = scale ? a * dis[r] : a;
      }
    }
  }
#endif
}

// coef[i] = V[i] . ws  for i in [0, nvec)  (wave per basis vector, lanes strided over elements)
DSS_DEV void basis_dots(const float* V, int ldv, int nvec, const float* ws, int N, float* coef) {
  for (int i = DSS_WAVE; i < nvec; i += DSS_NWAVES) {
    const float* v = V + (size_t)i * ldv;
    float acc = 0.f;
    for (int e = DSS_LANE; e < N; e += DSS_LANES) acc += v[e] * ws[e];
    acc = DSS_WAVE_SUM(acc);
    if (DSS_LANE == 0) coef[i] = acc;
  }
}

// ws[e] -= sum_i coef[i] * V[i][e]
DSS_DEV void basis_axpy(const float* V, int ldv, int nvec, float* ws, int N, const float* coef) {
  for (int e = DSS_TID; e < N; e += DSS_NT) {
    float acc = ws[e];
    for (int i = 0; i < nvec; ++i) acc -= coef[i] * V[(size_t)i * ldv + e];
    ws[e] = acc;
  }
}

// Symmetric eigen-decomposition of the (m x m) projected matrix by the classical two-sided Jacobi method with
// a parallel (round-robin) ordering, fp64, entirely in LDS.  A round rotates M/2 DISJOINT index pairs at once:
//   1. one thread per pair: (c, s) from a_pp, a_qq, a_pq                       - no reductions at all
//   2. one thread per (row, pair): columns p,q of A and of the accumulated V   (A <- A J, V <- V J)
//   3. one thread per (pair, column): rows p,q of A                            (A <- J^T A)
// with a barrier after each step.  A and Vr are column-major with leading dimension m (A is symmetric).
// On exit theta[c] = A[c][c], Vr[:, c] = eigenvector, perm = ranks (descending theta).
DSS_DEV void jacobi_eig(double* A, double* Vr, int m, EigsSmall* sm) {
  const int M = (m + 1) & ~1, np = M / 2;
  for (int sweep = 0; sweep < 30; ++sweep) {
    if (DSS_TID == 0) sm->flag = 0;
    DSS_SYNC();
    for (int round = 0; round < M - 1; ++round) {
      for (int k = DSS_TID; k < np; k += DSS_NT) {
        int p, q;
        if (k == 0) { p = M - 1; q = round; }
        else { p = (round + k) % (M - 1); q = (round - k + (M - 1)) % (M - 1); }
        if (p > q) { const int t = p; p = q; q = t; }
        double c = 1.0, sn = 0.0;
        if (q < m) {
          const double app = A[(size_t)p * m + p], aqq = A[(size_t)q * m + q], apq = A[(size_t)q * m + p];
          if (fabs(apq) > 1e-17 + 1e-15 * sqrt(fabs(app * aqq))) {
            const double zeta = (aqq - app) / (2.0 * apq);
            const double t = (zeta >= 0. ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            c = 1.0 / sqrt(1.0 + t * t);
            sn = c * t;
            sm->flag = 1;  // benign race: every writer stores 1
          }
        } else {
          q = p;  // dummy pair (odd m): identity
        }
        sm->jp[k] = p; sm->jq[k] = q; sm->jc[k] = c; sm->js[k] = sn;
      }
      DSS_SYNC();
      for (int idx = DSS_TID; idx < np * m; idx += DSS_NT) {  // columns p,q of A and V, one row each
        const int k = idx / m, r = idx - k * m;
        const int p = sm->jp[k], q = sm->jq[k];
        if (p == q) continue;
        const double c = sm->jc[k], sn = sm->js[k];
        const double ap = A[(size_t)p * m + r], aq = A[(size_t)q * m + r];
        A[(size_t)p * m + r] = c * ap - sn * aq;
        A[(size_t)q * m + r] = sn * ap + c * aq;
        const double vp = Vr[(size_t)p * m + r], vq = Vr[(size_t)q * m + r];
        Vr[(size_t)p * m + r] = c * vp - sn * vq;
        Vr[(size_t)q * m + r] = sn * vp + c * vq;
      }
      DSS_SYNC();
      for (int idx = DSS_TID; idx < np * m; idx += DSS_NT) {  // rows p,q of A, one column each
        const int k = idx / m, col = idx - k * m;
        const int p = sm->jp[k], q = sm->jq[k];
        if (p == q) continue;
        const double c = sm->jc[k], sn = sm->js[k];
        const double ap = A[(size_t)col * m + p], aq = A[(size_t)col * m + q];
        A[(size_t)col * m + p] = c * ap - sn * aq;
        A[(size_t)col * m + q] = sn * ap + c * aq;
      }
      DSS_SYNC();
    }
    if (sm->flag == 0) break;  // a full sweep without a rotation: converged (uniform: read after the barrier)
    DSS_SYNC();
  }
  for (int c = DSS_TID; c < m; c += DSS_NT) sm->theta[c] = A[(size_t)c * m + c];
  DSS_SYNC();
  for (int c = DSS_TID; c < m; c += DSS_NT) {
    int rank = 0;
    const double tc = sm->theta[c];
    for (int j = 0; j < m; ++j) {
      const double tj = sm->theta[j];
      rank += (tj > tc || (tj == tc && j < c)) ? 1 : 0;
    }
    sm->perm[rank] = c;
  }
  DSS_SYNC();
}

// Rayleigh-Ritz on the projected matrix T (m x m): diagonal alpha, arrow column l (after a restart),
// off-diagonal beta[j] = T[j][j+1] for j >= l.  Returns the number of the K wanted Ritz pairs whose residual
// |beta_last * z_{m-1,i}| exceeds tol * max(|theta_i|, 1e-3).  Leaves Vr / sm->theta / sm->perm set.
DSS_DEV int rayleigh_ritz(double* A, double* Vr, int m, int l, int K, double beta_last, float tol, EigsSmall* sm) {
  for (int idx = DSS_TID; idx < m * m; idx += DSS_NT) {
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
  DSS_SYNC();
  jacobi_eig(A, Vr, m, sm);
  int nbad = 0;
  for (int i = 0; i < K; ++i) {
    if (i >= m) { ++nbad; continue; }
    const int c = sm->perm[i];
    const double res = fabs(beta_last * Vr[(size_t)c * m + (m - 1)]);
    const double th = fabs(sm->theta[c]);
    if (res > (double)tol * (th > 1e-3 ? th : 1e-3)) ++nbad;
  }
  return nbad;
}

// The whole eigen stage for ONE image (called by every thread of the owning workgroup).
//   W          [N, ld] f32, symmetric, non-negative, pad columns zero
//   gws        global workspace of eigs_ws_floats_per_image(ld, ncv) floats
//   lds        LDS block of eigs_lds_layout(ld, ncv).total bytes (16-byte aligned)
//   eigenvalues[K], eigenvectors[K, N] outputs; *info = +passes (converged) / -passes (budget exhausted)
DSS_DEV void eigs_one_image(const float* __restrict__ W, const EigsParams P, float* gws, unsigned char* lds,
                            float* eigenvalues, float* eigenvectors, int32_t* info) {
  const int N = P.N, ld = P.ld, K = P.K, mmax = P.ncv;
  const EigsLds L = eigs_lds_layout(ld, mmax);
  float* xs = reinterpret_cast<float*>(lds + L.off_xs);
  float* ws = reinterpret_cast<float*>(lds + L.off_ws);
  double* A = reinterpret_cast<double*>(lds + L.off_A);
  double* Vr = reinterpret_cast<double*>(lds + L.off_V);
  EigsSmall* sm = reinterpret_cast<EigsSmall*>(lds + L.off_small);
  const int ldv = ld;
  float* Va = gws;
  float* Vb = gws + (size_t)(mmax + 1) * ldv;
  float* dis = gws + (size_t)2 * (mmax + 1) * ldv;
  int passes = 0;

  // ---- degree: d = W 1 ; clamp ; dis = d^-1/2 -------------------------------------------------
  for (int e = DSS_TID; e < ld; e += DSS_NT) xs[e] = e < N ? 1.0f : 0.0f;
  DSS_SYNC();
  matvec_rows(W, N, ld, xs, ws, nullptr, false);
  ++passes;
  DSS_SYNC();
  for (int e = DSS_TID; e < ld; e += DSS_NT) {
    float d = e < N ? ws[e] : 1.0f;
    if (d < 1e-12f) d = 1.0f;
    dis[e] = e < N ? 1.0f / sqrtf(d) : 0.0f;
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

  int l = 0;          // kept Ritz vectors (0 on the first cycle)
  int m = mmax;       // effective Krylov dimension of this cycle (shrinks on breakdown)
  double beta_last = 0.;
  bool converged = false;
  int restart = 0;
  for (;; ++restart) {
    // ---- extend the Krylov basis from l to m --------------------------------------------------------
    bool breakdown = false, early = false;
    for (int j = l; j < mmax; ++j) {
      const float* vj = Va + (size_t)j * ldv;
      for (int e = DSS_TID; e < ld; e += DSS_NT) xs[e] = e < N ? dis[e] * vj[e] : 0.0f;
      DSS_SYNC();
      matvec_rows(W, N, ld, xs, ws, dis, true);
      ++passes;
      DSS_SYNC();
      // classical Gram-Schmidt, two passes (full reorthogonalisation against V[0..j])
      basis_dots(Va, ldv, j + 1, ws, N, sm->coef);
      DSS_SYNC();
      double alpha = (double)sm->coef[j];
      basis_axpy(Va, ldv, j + 1, ws, N, sm->coef);
      DSS_SYNC();
      basis_dots(Va, ldv, j + 1, ws, N, sm->coef);
      DSS_SYNC();
      alpha += (double)sm->coef[j];
      basis_axpy(Va, ldv, j + 1, ws, N, sm->coef);
      DSS_SYNC();
      float b2 = 0.f;
      for (int e = DSS_TID; e < N; e += DSS_NT) b2 += ws[e] * ws[e];
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
      // mid-cycle convergence check every 2 steps: a converged image stops streaming W at once
      if (m < mmax && m >= K + 3 && m > l + 1 && ((m - l) & 1) == 0) {
        if (rayleigh_ritz(A, Vr, m, l, K, beta_last, P.tol, sm) == 0) { early = true; break; }
        DSS_SYNC();
      }
    }
    // ---- Rayleigh-Ritz on the full basis (skipped when a mid-cycle check already converged) -----------------
    int nbad = 0;
    if (!early) nbad = rayleigh_ritz(A, Vr, m, l, K, beta_last, P.tol, sm);
    converged = (nbad == 0);
    if (converged || breakdown || restart >= P.max_restarts) break;
    // ---- thick restart: keep the best `keep` Ritz vectors -----------------------------------------------
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
    { float* t = Va; Va = Vb; Vb = t; }
    l = keep;
    DSS_SYNC();
  }

  // ---- Ritz vectors -> generalized eigenvectors v = D^-1/2 u, sign rule, eigenvalues ----------------------
  DSS_SYNC();
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
      const float v = u * dis[e];
      ws[e] = v;
      pos += v > 0.f ? 1 : 0;
    }
    const int cnt = (int)(block_sum((float)pos, sm) + 0.5f);
    // extract/extract.py:238-240: negate iff 0.5 < mean(v > 0) < 1.0  <=>  2*cnt > N and cnt < N
    const float sgn = (2 * cnt > N && cnt < N) ? -1.0f : 1.0f;
    for (int e = DSS_TID; e < N; e += DSS_NT) eigenvectors[(size_t)i * N + e] = sgn * ws[e];
    DSS_SYNC();
  }
  for (int i = DSS_TID; i < K; i += DSS_NT) eigenvalues[i] = (float)(1.0 - sm->theta[sm->perm[i]]);
  if (DSS_TID == 0) *info = converged ? passes : -passes;
}

}  // namespace dss
