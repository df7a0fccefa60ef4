/* dss_hip.h - C ABI of libdss_hip.so: the MI355X (gfx950) kernels behind the
 * deep-spectral-segmentation `extract.py` hot path (extract_features -> extract_eigs).
 *
 * The reference has no FFI/plugin seam of its own (pure Python): its seam is
 * the two CLI commands, their Python signatures and two .pth schemas.  This header is the
 * native boundary underneath that seam.  Every entry point cites the reference lines whose
 * arithmetic it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no C++/torch types.  All pointers are DEVICE pointers
 *     (HBM) owned by the caller unless stated; row-major; fp32 unless a dtype argument says so.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Every call
 *     only ENQUEUES work on that stream and returns; nothing synchronises, nothing allocates.
 *   - Return value: DSS_OK (0) or a negative DSS_ERR_* code; dss_last_error() gives the message
 *     (thread-local).  Functions are re-entrant; no environment variable is read; the only global state is dss_linear_lt's (see there).
 *   - Half-precision dtypes: DSS_F16 (IEEE binary16) / DSS_BF16; DSS_F32 where noted.
 */
#ifndef DSS_HIP_H
#define DSS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 10: + dss_linear_lt_accumulate (the Mlp branch added to the fp32 residual stream inside the GEMM's epilogue);
 * 9: + dss_linear_lt / _workspace_bytes / _describe (round 6: the library GEMMs behind this ABI, never a Stream-K algorithm);
 * 8: the packed storage of W gained the EDGE STRIP (below, at dss_affinity): a caller that only passes W from dss_affinity* to
 * dss_*_eigs* - every caller there is - is unaffected; one that builds or reads packed W itself must follow the layout of the
 * library it runs against (dss_affinity_elems(N) tells them apart: 106 * 4096 at N = 900 with the strip, 120 * 4096 without);
 * 7: + dss_lnlinear_kfeatures (D = 384 / 768, f16 / bf16 operands; round 5);  6: + dss_patch_embed_p16;  5: + dss_lnlinear_kfeatures_k384;  4: + dss_lnlinear_prepare / _k384 / _k768 (round 4).  Entry points are
 * only ever added: a caller built against version n runs against any library with dss_abi_version() >= n. */
#define DSS_ABI_VERSION 10

enum { DSS_F32 = 0, DSS_F16 = 1, DSS_BF16 = 2 };

/* Layout of a [rows, D] half-precision activation (D % 64 == 0 for DSS_PLANAR64):
 *   DSS_ROW_MAJOR : element (r, c) at r * D + c                       - what torch.nn.Linear produces;
 *   DSS_PLANAR64  : element (r, c) at (c / 64) * rows * 64 + r * 64 + c % 64, i.e. [D/64][rows][64] - every
 *                   64-column group (one attention head) is a plane of contiguous 128-byte rows.  Written by
 *                   dss_linear_k384, read by dss_attention_fwd (qkv) and dss_layernorm_fwd (residual): a wave's
 *                   64 x 64 output tile is one contiguous 8 KB run instead of 64 scattered 128-byte lines. */
enum { DSS_ROW_MAJOR = 0, DSS_PLANAR64 = 1 };

enum {
  DSS_OK = 0,
  DSS_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, unsupported shape/dtype */
  DSS_ERR_HIP = -2,          /* a HIP runtime call / kernel launch failed                */
  DSS_ERR_WORKSPACE = -3,    /* workspace too small                                      */
  DSS_ERR_NO_CONVERGENCE = -4 /* reported per image through `info`, never as return code  */
};

/* ---- library ------------------------------------------------------------------------- */
int dss_abi_version(void);
const char* dss_last_error(void);
/* Name of the gfx target the kernels were compiled for ("gfx950"). */
const char* dss_target_arch(void);

/* ---- a2/a3/a5: image transform ---------------------------------------------------------
 * extract/extract_utils.py:55-56  ToTensor (u8 HWC -> f32 CHW, /255) + Normalize(ImageNet)
 * extract/extract.py:82-88        crop to (H//P*P, W//P*P), top-left
 * img_u8: [B, H, W, 3] RGB.  out_chw: [B, 3, H, W] f32 - the uncropped transformed image,
 * bit-exact to the reference transform (IEEE division, same operation order). */
int dss_preprocess_chw(const uint8_t* img_u8, float* out_chw, int B, int H, int W, void* stream);
/* Fused transform + crop + im2col for the patch-embedding GEMM:
 * out: [B, (H/P)*(W/P), 3*P*P] in `out_dtype`, inner index (c, py, px) = Conv2d weight order;
 * patch index n = (y/P)*(W/P) + (x/P)  (row-major, the reference's patch order). */
int dss_preprocess_patchify(const uint8_t* img_u8, void* out, int B, int H, int W, int P,
                            int out_dtype, void* stream);

/* ---- a6': LayerNorm (DINO blocks' norm1/norm2, eps 1e-6) ------------
 * y[r,:] = (x[r,:] - mean) * rsqrt(var + eps) * gamma + beta, statistics in fp32 (biased var).
 * x: [rows, D] f32.  y: [rows, D] in out_dtype.  D % 4 == 0, D <= 2048.
 * If `residual` != NULL (dtype `res_dtype`, [rows, D]) the kernel first does x += residual
 * IN PLACE (the block's `x = x + attn(...)` / `x = x + mlp(...)`) and normalises the sum. */
int dss_layernorm_fwd(float* x, const void* residual, int res_dtype, int res_layout, const float* gamma,
                      const float* beta, void* y, int out_dtype, int rows, int D, float eps,
                      void* stream);

/* ---- a6'': multi-head self-attention (DINO Attention.forward, heads of 64) ----------------
 * qkv: [B, T, 3, heads, 64] in `dtype` (the qkv Linear's output, untouched) when qkv_layout == DSS_ROW_MAJOR, or
 *      the same [B*T, 3*heads*64] matrix in DSS_PLANAR64 layout (plane index = which * heads + head);
 * out: [B, T, heads*64] in `dtype`  = softmax(q k^T * scale) v, heads re-interleaved as the
 * reference's `.transpose(1, 2).reshape(B, T, C)`.  fp32 accumulation and softmax statistics.
 * qkv is read in place (no workspace); one kernel: 8 waves x 32 queries per workgroup, four waves per SIMD (attention.hip). */
int dss_attention_fwd(const void* qkv, int qkv_layout, void* out, int B, int T, int heads, float scale, int dtype,
                      void* stream);

/* ---- a6: Linear layers whose reduction dimension is the embedding width (qkv / attn.proj / mlp.fc1 of DINO's Block)
 * C[M, N] = A[M, K] . W[N, K]^T + bias[N], optionally followed by the exact (erf) GELU of DINO's Mlp
 * (replaces torch.nn.Linear / nn.GELU inside DINO's Block; reached from extract/extract.py:94).
 * K = 384 (dino_vits16 / dino_vits8): dss_linear_k384, N <= 2048;  K = 768 (dino_vitb16 / dino_vitb8):
 * dss_linear_k768, N <= 3072.  A (row-major), W, bias, C in `dtype` (DSS_F16 / DSS_BF16), fp32 accumulation;
 * N % 64 == 0.  out_layout: DSS_ROW_MAJOR or DSS_PLANAR64 (see above).  K-resident MFMA kernel (linear384.hip).
 * gelu: 0 = none;  1 = erf-GELU evaluated in fp32 (A&S 7.1.28, |error| <= 3e-7: the output is the correctly rounded `dtype`
 *       value);  2 (round 5; DSS_F16 only) = the same function as a polynomial form on packed f16 - 5.5 instructions per value
 *       instead of 17.5, and they co-issue with the other wave's MFMAs: max |error| 1.1e-3 (0.6 of the output's f16 spacing
 *       where it occurs, at most 2.1 spacings for 0.25 < x < 0.5), <= 3.2e-4 for x < 0, relative <= 2e-3 for |x| < 0.5; its
 *       arithmetic is restated in tests/util.gelu_f16_poly and pinned bit for bit on the GPU. */
int dss_linear_k384(const void* A, const void* W, const void* bias, void* C, int M, int N, int gelu, int out_layout,
                    int dtype, void* stream);
int dss_linear_k768(const void* A, const void* W, const void* bias, void* C, int M, int N, int gelu, int out_layout,
                    int dtype, void* stream);

/* ---- a6: the Linear layers that are not hand-written kernels (mlp.fc2; at D = 768 also attn.proj and, at patch size 8, the
 * patch embedding) - torch.nn.Linear inside DINO's Block / PatchEmbed (reached from extract/extract.py:94):
 *     C[M, N] = A[M, K] . W[N, K]^T + bias[N]        (bias may be NULL)
 * as a hipBLASLt GEMM that is verified to run WITHOUT Stream-K's partial-tile exchange (gemm.hip; round 6).  Every gfx950 kernel
 * of this stack's hipBLASLt splits the last, partly filled round of output tiles across workgroups through a workspace, and that
 * exchange is not reproducible here (~1 launch in 56 000 returns different values in whole 256-row tiles:
 * profiles/r06_forward_stress.txt).  The library sets Tensile's switch TENSILE_STREAMK_DATA_PARALLEL=1 for the process before
 * it creates its hipblasLt handle (no overwrite) and takes, in the heuristic's own order, the first candidate for which hipBLASLt
 * then reports a workspace of 0 bytes - every output tile written by one workgroup, the same bits on every launch; if no
 * candidate does (hipBLASLt initialised earlier in the process without the switch), the call FAILS instead of running a split
 * kernel.  A, W, bias in `dtype` (DSS_F16 / DSS_BF16), row-major;
 * C in `out_dtype` = `dtype` or DSS_F32 (the K projection keeps its fp32 accumulators); fp32 accumulation.  `workspace`: at least
 * dss_linear_lt_workspace_bytes() bytes the caller owns (the upper bound a candidate may ask for; the ones taken ask for none).
 * The choice is cached per
 * (M, N, K, dtypes, bias): the one hipblasLt handle and that cache are the library's only persistent state (mutex-guarded).
 * Among the qualifying candidates the first is taken, unless a MEASURED preference names a tile shape for the problem class and a
 * qualifying candidate has it (one entry so far: N = 384, K = 1536, f16, from 65 536 rows - profiles/r06_lt_tune.txt).
 * dss_linear_lt_describe writes the candidate list for a problem into buf, one line per candidate ('* ' = the one taken, '*p' = taken
 * by a measured preference, 'x' = passed over: a partial-tile workspace, single-buffer split-K, or a workspace beyond workspace_bytes). */
size_t dss_linear_lt_workspace_bytes(void);
int dss_linear_lt(const void* A, const void* W, const void* bias, void* C, long M, int N, int K, int dtype, int out_dtype,
                  void* workspace, size_t workspace_bytes, void* stream);
int dss_linear_lt_describe(long M, int N, int K, int dtype, int out_dtype, int has_bias, size_t workspace_bytes, char* buf,
                           size_t buflen);
/* The same GEMM ADDED to the fp32 residual stream in place - DINO Block's `x = x + mlp(...)` with the add inside the GEMM's own
 * epilogue (extract/extract.py:94 -> Block.forward):   X[M, N] (f32) += A[M, K] . W[N, K]^T + bias[N]
 * (hipBLASLt's beta = 1 with C = D = X: the branch output is added from the fp32 accumulators, never rounded to `dtype`, and the
 * LayerNorm kernel behind it reads the finished stream - no pending branch output to add, no x write-back there).  `bias` is
 * fp32 like the stream (or NULL).  ABI 10. */
int dss_linear_lt_accumulate(const void* A, const void* W, const float* bias, float* X, long M, int N, int K, int dtype,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- a6 + a6': residual add + LayerNorm + Linear in ONE kernel (DINO Block: `x = x + branch; h = norm(x); y = lin(h)`,
 * i.e. norm1 -> attn.qkv and norm2 -> mlp.fc1 (+ GELU); SURVEY.md Appendix A, reached from extract/extract.py:94).
 *   x        [M, K] f32, the residual stream, UPDATED IN PLACE: x += residual (skipped when residual == NULL);
 *   residual [M, K] in `dtype`, layout res_layout (DSS_ROW_MAJOR or DSS_PLANAR64), or NULL;
 *   C        = act( LayerNorm_eps(x) . W^T + bias ) with LayerNorm's gamma/beta, W and bias given in the folded form that
 *              dss_lnlinear_prepare builds once per layer:  Wg[n][k] = dtype(W[n][k] gamma[k]),
 *              aux[n] = (-sum_k Wg[n][k],  bias[n] + sum_k W[n][k] beta[k])  (f32 [N, 2]).
 * The kernel keeps dtype(x - pivot) as its K-resident operand (ONE rounding; pivot = a robust typical value of the row - the
 * median of the medians of three column triples -, so the rounding error scales with the row's spread, not with |x|: rows whose
 * mean is far from zero, up to beyond the f16 range, are as accurate as with a standalone LayerNorm pass; round 5) and applies
 * mean / sigma exactly: C = rstd * (xp . Wg^T - (mean - pivot) * sum_k Wg + sigma * b'), the two corrections as one fp32 MFMA
 * step.  Statistics in fp32 from pivot-shifted moments (biased variance, like torch.nn.LayerNorm).  Same shapes / layouts / dtypes as dss_linear_k384
 * / _k768 (K = 384 / 768).  x, residual and C must not alias. */
int dss_lnlinear_prepare(const float* W, const float* bias, const float* gamma, const float* beta, void* Wg, float* aux,
                         int N, int K, int dtype, void* stream);
int dss_lnlinear_k384(float* x, const void* residual, int res_layout, float eps, const void* Wg, const float* aux, void* C,
                      int M, int N, int gelu, int out_layout, int dtype, void* stream);
int dss_lnlinear_k768(float* x, const void* residual, int res_layout, float eps, const void* Wg, const float* aux, void* C,
                      int M, int N, int gelu, int out_layout, int dtype, void* stream);

/* ---- a6'' + a8 + a9: the LAST hooked block's K projection, from the residual stream to what the caller and the affinity
 * build take over (extract/extract.py:89-98: the forward hook on blocks[which_block].attn.qkv, `k` of its output, CLS
 * dropped; :148 F.normalize's norms).  dss_lnlinear_k384 with the K rows of the qkv weight (N = K = 384, folded by
 * dss_lnlinear_prepare) whose epilogue writes, for token row b * T + t with t >= 1, output row b * (T - 1) + t - 1 of
 *   k32   [M / T * (T - 1), 384] f32  the features (fp32 accumulators, never rounded),
 *   k16   the same rounded to f16, and
 *   rnorm [M / T * (T - 1)]  = 1 / max(|k16 row|_2, norm_eps)   (the norm of the ROUNDED row: w_ii = 1 exactly);
 * CLS rows are computed and dropped.  x [M, 384] f32 is READ ONLY here (the hooked block is the stream's last reader: x + residual
 * is used and not stored - the one difference from dss_lnlinear_k384's prologue).
 * f16 operands only; 64 < T, M * T < 2^32.  Replaces dss_layernorm_fwd + a library GEMM + dss_kfeatures_finalize. */
int dss_lnlinear_kfeatures_k384(float* x, const void* residual, int res_layout, float eps, const void* Wg, const float* aux,
                                float* k32, void* k16, float* rnorm, int M, int T, float norm_eps, void* stream);
/* The same hand-over for D = 384 or 768 (dss_lnlinear_k768's body at one row tile per wave) and f16 or bf16 operands (Wg and the
 * residual are `dtype`; k16 is ALWAYS f16 - what dss_affinity_f16_u16 reads).  M * D * 4 < 2^32.  ABI v7 (round 5): the
 * D = 768 models' (dino_vitb8 / vitb16) last LayerNorm launch, K-projection GEMM and finalize pass in one kernel. */
int dss_lnlinear_kfeatures(float* x, const void* residual, int res_layout, float eps, const void* Wg, const float* aux,
                           float* k32, void* k16, float* rnorm, int M, int T, int D, float norm_eps, int dtype, void* stream);

/* ---- a3 + a5 + a6 (first step): ToTensor + Normalize, the crop to whole patches, DINO's PatchEmbed Conv2d(3, D, 16, 16) and
 * `x = cat(cls, tokens) + pos_embed` for the patch rows, in ONE kernel from the u8 image (extract/extract_utils.py:55-56,
 * extract/extract.py:82-88).  img_u8 [B, H, W, 3]; Wp [D, 768] and biasp [D] in `dtype` are the caller's
 * FOLDED parameters: Wp[n][(py, px, c)] = W[n][c][py][px] / (255 std_c), biasp[n] = b[n] - sum W[n][c][py][px] (mean_c - 128 / 255) /
 * std_c, so that the operand is pixel - 128 (-128..127: exact in f16 and bf16, and centred, so that the rounding of Wp scales a
 * deviation and not the 0..255 level); pos [Np, D] f32 = the (interpolated) position embedding
 * of the patches (a caller that wants the folded bias - several times the conv's own - unrounded adds it to these rows and
 * passes biasp = zeros: that is what the package does); x [B, Np + 1, D] f32: rows 1..Np of every image are written (row 0, the CLS token, is the caller's).
 * Patch size 16 only (K = 3 * 16 * 16 = 768: the K-resident kernel of linear384.hip gathers a patch as its operand row);
 * D % 64 == 0, D <= 3072.  Replaces dss_preprocess_patchify + a GEMM + an elementwise pass. */
int dss_patch_embed_p16(const uint8_t* img_u8, const void* Wp, const void* biasp, const float* pos, float* x, int B, int H, int W,
                        int D, int dtype, void* stream);

/* ---- a10: row L2 normalisation -------------------------------------------------------------
 * extract/extract.py:148  F.normalize(feats, p=2, dim=-1):  y = x / max(||x||_2, eps). */
int dss_normalize_rows(const float* x, float* y, int rows, int D, float eps, void* stream);

/* ---- a12: patch-feature affinity -------------------------------------------------------------
 * extract/extract.py:191-194  W = F F^T ; W = W * (W > 0)   (exact fp32 MFMA, fmaf-chain numerics).
 * `W / W.max()` (:194) is NOT applied: the generalized problem (D-W)v = lambda D v is invariant
 * under W -> cW; eigenvalues and eigenvectors are unchanged.
 * feats: [B, N, D] f32 (already normalised if wanted).
 * W is symmetric, so it is produced (and later streamed) as PACKED UPPER-TRIANGULAR STORAGE, in blocks of 4096 elements.  With
 * ld = dss_affinity_ld(N) = N rounded up to 64 and nt = ld/64 tile rows:
 *   - no edge strip (N mod 64 == 0 or > 16, or N <= 64): ntf = nt, and tile (I, J), I <= J < ntf, is block
 *     t = I*ntf - I*(I-1)/2 + (J-I), row-major inside the 64x64 tile, diagonal tiles stored in full;
 *   - EDGE STRIP (1 <= N mod 64 <= 16, N > 64; ABI 8): ntf = nt - 1 full tile rows / columns as above, and of the last tile
 *     column only E = 4 (N mod 64 <= 4) or 16 columns are kept, as MINI TILES of 64 rows x 4 columns ([row][4], 256 elements):
 *     mini tile m = I * (E/4) + e holds columns 64 ntf + 4e .. + 3 of rows 64 I .. 64 I + 63, I = 0 .. ntf (I = ntf is the
 *     corner, stored in full); they follow the full tiles, the last block of 16 mini tiles padded (never read).
 *     N = 900: 105 + 1 blocks instead of 120 (1.07x the N (N + 1) / 2 elements of the triangle instead of 1.21x).
 * Entries with row or column >= N are 0.  Per image dss_affinity_elems(N) elements: W is [B, dss_affinity_elems(N)].
 * (csrc/eigs_core.h: wsym_layout / wsym_at; dss_amd.hip.affinity_to_dense / affinity_from_dense convert.) */
int dss_affinity_ld(int N);
size_t dss_affinity_elems(int N);
int dss_affinity(const float* feats, float* W, int B, int N, int D, int threshold_at_zero,
                 void* stream);

/* Same W (same packed layout), fused with the row normalisation, at the f16 MFMA rate with fp32-class accuracy:
 * each normalised feature is split x = hi + lo (two f16 values) and <x_i,x_j> = hi.hi + hi.lo + lo.hi is
 * accumulated in fp32 (error ~1e-7; the fp32 kernel above is bitwise an fmaf chain).  This moves the affinity
 * build from the fp32-MFMA roofline to the HBM roofline.  feats: RAW features [B, N, D] f32; normalize != 0
 * applies extract.py:148 (eps as in dss_normalize_rows).  workspace: dss_affinity_split_workspace_bytes(B, N, D). */
size_t dss_affinity_split_workspace_bytes(int B, int N, int D);
int dss_affinity_split(const float* feats, float* W, int B, int N, int D, int normalize, float eps,
                       int threshold_at_zero, void* workspace, size_t workspace_bytes, void* stream);
/* The default recipe (normalize = True, threshold_at_zero = True) with W stored as 16-bit fixed point:
 * W_q = round(65535 w), w in [0, 1], same packed tile layout, dss_affinity_elems(N) uint16 per image - half the
 * bytes of the only HBM stream of the eigensolver.  The normalised-Laplacian eigenproblem does not depend on the scale
 * of W; the uniform 7.6e-6 quantisation step moves the eigenvectors of the reference goldens by <= 1e-6 in cosine. */
int dss_affinity_split_u16(const float* feats, uint16_t* W, int B, int N, int D, float eps, void* workspace,
                           size_t workspace_bytes, void* stream);

/* The default recipe again, as ONE kernel with HBM traffic == algorithmic bytes (4 N D in, 2 * dss_affinity_elems(N) out
 * per image; no workspace): raw fp32 features in, normalisation applied after the product (the kernel collects the squared
 * row norms while it streams the panels), f16 MFMA operands with fp32 accumulation (BASELINE.json config 5: "fp16 features
 * + fp32 Laplacian eigensolve"), W_q = round(65535 relu(w)) out.  |W - W_exact| <= ~5e-5; the eigenvectors of the reference
 * goldens move by <= 3e-6 in cosine.  D % 32 == 0.  extract/extract.py:148,191-193. */
int dss_affinity_fused_u16(const float* feats, uint16_t* W, int B, int N, int D, float eps, void* stream);

/* The same build for features that never were fp32 on the way in - the in-memory pipeline, where the K projection hands
 * its output over through dss_kfeatures_finalize: f16 features [B, N, D] + their inverse norms [B, N] in, packed 16-bit W
 * out.  256 x 128 block tiles, panels by LDS-DMA (affinity.hip: gram_f16_dma_kernel).  Algorithmic bytes per image:
 * 2 N D + 4 N in, 2 * dss_affinity_elems(N) out.  D % 32 == 0.  extract/extract.py:148,191-193. */
int dss_affinity_f16_u16(const void* feats16, const float* rnorm, uint16_t* W, int B, int N, int D, void* stream);

/* a7/a10 hand-over between the two stages when they run back to back in HBM (extract/extract.py:96-98,148): the raw fp32
 * output of the last block's K projection kproj [B, T, D] (token 0 = CLS; + bias [D] unless NULL) -> k32 [B, T-1, D] fp32
 * (the hooked features, CLS dropped: what extract_features saves), k16 the same rounded to f16, rnorm [B, T-1] =
 * 1 / max(|k16 row|, eps).  One pass; replaces a bias add and a strided copy. */
int dss_kfeatures_finalize(const float* kproj, const float* bias, float* k32, void* k16, float* rnorm, int B, int T, int D,
                           float eps, void* stream);

/* ---- a13-a15: degree, normalised Laplacian, K smallest generalized eigenpairs, sign rule -------
 * extract/extract_utils.py:207-220  d = W 1 ; d[d < 1e-12] = 1
 * extract/extract.py:227            eigsh(D - W, k=K, sigma=0, which='LM', M=D)
 * extract/extract.py:235-240        eigenvectors.T (f32 [K, N]) ; sign rule
 * Solved as the K LARGEST eigenpairs (mu, u) of S = D^-1/2 W D^-1/2 by thick-restart Lanczos
 * with full reorthogonalisation (one workgroup per image; the stored half of W streamed once per Lanczos step);
 * lambda = 1 - mu ascending, v = D^-1/2 u  (so v^T D v = 1, the reference's normalisation).
 * W: [B, dss_affinity_elems(N)] f32, packed as dss_affinity writes it.  eigenvalues: [B, K] f32.
 * eigenvectors: [B, K, N] f32.
 * info: [B] int32 - number of W passes (>0) if converged, -(passes) if the restart budget ran out
 * (outputs then hold the best available Ritz pairs).
 * ncv: Krylov dimension (0 = default max(2K+10, 20), capped at 64); tol: Ritz residual tolerance
 * relative to max(|mu|, 1e-3) (0 = default 2e-6); max_restarts (0 = default 60). */
size_t dss_eigs_workspace_bytes(int B, int N, int K, int ncv);
int dss_laplacian_eigs(const float* W, int B, int N, int K, float* eigenvalues, float* eigenvectors,
                       int32_t* info, int ncv, float tol, int max_restarts,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Same solver on the W written by dss_affinity_split_u16 (outputs in the reference's units: v^T D v = 1 with the
 * true degrees). */
int dss_laplacian_eigs_u16(const uint16_t* W, int B, int N, int K, float* eigenvalues, float* eigenvectors,
                           int32_t* info, int ncv, float tol, int max_restarts,
                           void* workspace, size_t workspace_bytes, void* stream);

/* The same solver on the other two problems of the reference's _extract_eig (same W layout, same outputs' shapes,
 * pairs returned in the solver's ranking order; the sign rule is applied to every vector):
 *   DSS_EIGS_NORMALIZED_LAPLACIAN (0)  = dss_laplacian_eigs                                   extract.py:227
 *   DSS_EIGS_AFFINITY_LM          (1)  K eigenpairs of W of largest |value|, descending |value|  extract.py:171
 *                                      (with W = F F^T un-thresholded: squared singular values / left singular
 *                                       vectors of F, extract.py:161-163)
 *   DSS_EIGS_LAPLACIAN            (2)  K smallest eigenpairs of D - W (lapnorm=False), ascending  extract.py:232 */
enum { DSS_EIGS_NORMALIZED_LAPLACIAN = 0, DSS_EIGS_AFFINITY_LM = 1, DSS_EIGS_LAPLACIAN = 2 };
int dss_symmetric_eigs(const float* W, int B, int N, int K, int mode, float* eigenvalues, float* eigenvectors,
                       int32_t* info, int ncv, float tol, int max_restarts,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- a15: sign rule alone (extract/extract.py:238-240), in place on [rows, N] ----------------- */
int dss_sign_rule(float* eigenvectors, int rows, int N, void* stream);

/* ---- f1: the immediate consumers of the eigenvectors, on the device right after the solve (no .pth round trip) ------
 * dss_fiedler_mask: extract/extract.py:383-407 - mask[b][e] = eigenvectors[b][index][e] > threshold ? 255 : 0
 * (eigenvectors [B, K, N] f32 as the solvers write them; mask [B, N] u8: the 8-bit PNG the reference saves, row-major
 * over the patch grid).
 * dss_kmeans_segments: extract/extract.py:283-352 - K-means over the N points whose coordinates are eigenvectors
 * [first, first + dims) of each image (the reference clusters `eigenvectors[1:1+num_eigenvectors].T`), Lloyd iterations
 * with sklearn's two stopping rules (labels unchanged; squared centre shift <= tol * mean coordinate variance) and final
 * assignment; centroids_init [B, k, dims] or NULL (k-means++ seeding from a counter-based generator: seed, image, draw);
 * then, if infer_bg, the border vote of extract_utils.py:124-135 on the hp x wp grid (hp * wp == N; corners count
 * twice) and the label swap that makes the winning segment 0.  labels [B, N] u8, inertia [B] f32 (sum of squared
 * distances to the final centres), iters [B].  Limits: N <= 8192, dims <= 64, k <= 32.  An empty cluster keeps its
 * centre (sklearn relocates it).  The CLI's extract_multi_region_segmentations keeps clustering with sklearn on the host,
 * bit-identical to the reference for a seeded run; this entry point is the same algorithm for device-resident pipelines. */
int dss_fiedler_mask(const float* eigenvectors, uint8_t* mask, int B, int K, int N, int index, float threshold,
                     void* stream);
int dss_kmeans_segments(const float* eigenvectors, int B, int K, int N, int first, int dims, int k,
                        const float* centroids_init, unsigned seed, int max_iter, float tol, int hp, int wp, int infer_bg,
                        uint8_t* labels, float* inertia, int32_t* iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSS_HIP_H */
