"""Per-image spectral step on the GPU: features -> affinity -> Laplacian eigenpairs.

Host-side orchestration of the HIP pipeline that replaces the body of the reference's ``_extract_eig``
for ``which_matrix in ('laplacian', 'matting_laplacian')`` with ``image_color_lambda == 0``
(extract/extract.py:146-148, 175-195, 215-240).  Everything stays in HBM: there is no N x N
device->host copy and no scipy.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import hip


class EigsNotConverged(RuntimeError):
    pass


_PROBLEM_MODE = {"laplacian": hip.EIGS_NORMALIZED_LAPLACIAN, "laplacian_unnormalized": hip.EIGS_LAPLACIAN,
                 "affinity": hip.EIGS_AFFINITY_LM, "affinity_svd": hip.EIGS_AFFINITY_LM}


def _reference_order(problem: str, ev: torch.Tensor, vec: torch.Tensor):
    """Arrange the solver's (ranking-ordered) pairs the way the reference saves them for this branch."""
    if problem == "affinity":  # eigsh's ascending values as they are, vectors flipped to descending (extract.py:171-172)
        order = torch.argsort(ev, dim=1)
        vorder = order.flip(1)
        return torch.gather(ev, 1, order), torch.gather(vec, 1, vorder[:, :, None].expand_as(vec))
    if problem == "affinity_svd":  # singular values descending = sqrt of the eigenvalues of F F^T (extract.py:161-163)
        order = torch.argsort(ev, dim=1, descending=True)
        return (torch.gather(ev, 1, order).clamp_min(0).sqrt(),
                torch.gather(vec, 1, order[:, :, None].expand_as(vec)))
    return ev, vec


MAX_LANCZOS_K = 62   # dss_symmetric_eigs: Krylov dimension <= 64 and ncv >= K + 2 (csrc/eigs_core.h EIGS_MAX_NCV)


def _dense_eigs_w(w: torch.Tensor, K: int, problem: str):
    """fp64 dense solve of one symmetric ``[N, N]`` affinity: pairs in the kernel's ranking order."""
    if problem in ("affinity", "affinity_svd"):
        th, u = torch.linalg.eigh(w)
        order = th.abs().argsort(descending=True)[:K]
        return th[order], u[:, order].T
    d = w.sum(1)
    d = torch.where(d < 1e-12, torch.ones_like(d), d)
    if problem == "laplacian":
        dis = d.rsqrt()
        th, u = torch.linalg.eigh(w * dis[:, None] * dis[None, :])
        return (1.0 - th).flip(0)[:K], (u * dis[:, None]).T.flip(0)[:K]
    lam, u = torch.linalg.eigh(torch.diag(d) - w)
    return lam[:K], u[:, :K].T


@torch.no_grad()
def dense_eigs(feats: torch.Tensor, K: int, normalize: bool, threshold_at_zero: bool, problem: str):
    """Exceptional path, never the hot one: the same eigenproblems solved densely in fp64 on the GPU
    (``torch.linalg.eigh``, rocSOLVER) image by image.  Used for (a) an image whose Lanczos run exhausted even the
    enlarged restart budget - the reference reacts to an ARPACK failure with a second solve too (``which='SM'``,
    extract.py:228-229) and ALWAYS writes a file - and (b) ``K > 62``, beyond the Krylov space of the HIP kernel (the
    reference accepts any ``K < N``).  Returns pairs in the kernel's ranking order, sign rule applied."""
    evs, vecs = [], []
    for f in feats:
        x = f.double()
        if normalize:
            x = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        w = x @ x.T
        if threshold_at_zero:
            w = w.clamp_min(0)
        ev, v = _dense_eigs_w(w, K, problem)
        evs.append(ev.float())
        vecs.append(v.float().contiguous())
    ev, vec = torch.stack(evs), torch.stack(vecs).contiguous()
    hip.sign_rule_(vec)
    return ev, vec


def reference_scale(problem: str, wmax: torch.Tensor, ev: torch.Tensor, vec: torch.Tensor):
    """Outputs of the eigenproblem on W -> outputs of the reference's problem on ``W / wmax`` (extract.py:194), per image.
    ``laplacian``: ``(D - W) v = lam D v`` is scale invariant, but ARPACK normalises ``v^T (D / wmax) v = 1``: vectors
    grow by ``sqrt(wmax)``.  ``laplacian_unnormalized``: the eigenvalues of ``(D - W) / wmax`` are ``lam / wmax``, unit
    eigenvectors unchanged.  An all-zero feature matrix (wmax = 0: NaN in the reference) is left unscaled."""
    wmax = torch.where(wmax > 0, wmax, torch.ones_like(wmax)).to(ev.dtype)
    if problem == "laplacian":
        return ev, vec * wmax.sqrt()[:, None, None]
    if problem == "laplacian_unnormalized":
        return ev / wmax[:, None], vec
    return ev, vec


@torch.no_grad()
def laplacian_eigs_from_features(feats: torch.Tensor, K: int, normalize: bool = True,
                                 threshold_at_zero: bool = True, ncv: int = 0, tol: float = 0.0,
                                 max_restarts: int = 0, max_bytes: int = 24 << 30, strict: bool = True,
                                 affinity_mode: str = "split", retry: bool = True,
                                 problem: str = "laplacian",
                                 upsample: Optional[Tuple[Tuple[int, int], Tuple[int, int]]] = None,
                                 w_dtype: str = "u16",
                                 feats16: Optional[torch.Tensor] = None, rnorm: Optional[torch.Tensor] = None
                                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``feats``: f32 ``[B, N, D]`` on the GPU (one row per patch).  Returns
    ``(eigenvalues [B, K], eigenvectors [B, K, N], info [B])``, all on the GPU.

    * eigenvalues ascending (``lambda_0 ~ 0``), eigenvectors D-orthonormal, sign rule applied - the
      conventions of the reference's ``.pth`` schema.
    * images are processed in chunks whose affinity matrices fit in ``max_bytes`` of HBM.
    * ``problem``: ``"laplacian"`` (default: generalized ``(D-W)v = lambda D v``, extract.py:227),
      ``"laplacian_unnormalized"`` (``lapnorm=False``, :232), ``"affinity"`` (largest-magnitude eigenpairs of W,
      :166-172: eigenvalues ascending, eigenvectors in DESCENDING order - the reference's own quirk) or
      ``"affinity_svd"`` (:160-163: top-K singular values / left singular vectors of the features, never thresholded).
    * ``upsample``: ``((H_patch, W_patch), (H_lr, W_lr))`` - the reference's feature upsampling when
      ``image_downsample_factor != patch_size`` (extract.py:179-188): the (already normalised) features are resized
      bilinearly (``align_corners=False``) from the patch grid to the low-resolution pixel grid before the affinity.
    * ``affinity_mode``: ``"split"`` (default) builds W with two-term split-f16 MFMAs (error ~1e-7:
      what ``extract_eigs`` needs for fp32 features read from ``.pth`` files - eigenvalues within 1e-5 of the reference);
      ``"fp32"`` uses exact fp32 MFMAs; ``"fused"`` = for the default recipe with 16-bit W and ``D >= 256``, ONE kernel from
      raw features to packed W with f16 MFMA operands (``|dW| <= ~5e-5``, eigenvalues within ~6e-5, eigenvectors within
      ~1e-5 in cosine) - free when the features come out of the half-precision ViT, which is where ``pipeline`` and
      ``bench.py`` use it - and the split build otherwise.  With the default recipe (``problem="laplacian"``,
      ``normalize``, ``threshold_at_zero``) W is stored as ``round(65535 w)`` in 16 bits (``w_dtype="f32"``
      keeps floats): the problem is scale-invariant and the eigenvectors move by <= 1e-6 in cosine.
    * ``feats16`` / ``rnorm``: the f16 copy of ``feats`` and its inverse row norms (``hip.kfeatures_finalize``: the
      hand-over of a ViT that ran just before, ``DinoViT.extract_k_f16``).  With them the ``"fused"`` build of the
      default recipe starts from the f16 rows (``hip.affinity_f16_u16``: half the bytes through the L2, panels by
      LDS-DMA) - the same arithmetic as ``hip.affinity_fused_u16``, which rounds to f16 itself.
    * ``retry``: images that exhaust their restart budget are re-solved once with the largest Krylov space.
      Checking for them reads ``info`` back (one device->host sync per call): throughput loops that must keep
      the host running ahead pass ``retry=False, strict=False`` and inspect ``info`` once at the end.
    * ``strict``: raise ``EigsNotConverged`` if an image is still unconverged after that (the reference would
      have raised ``ArpackNoConvergence`` into a bare ``except``)."""
    if feats.dim() == 2:
        feats = feats[None]
    assert feats.dim() == 3 and feats.dtype == torch.float32
    if affinity_mode not in ("fused", "split", "fp32"):
        raise ValueError(f"affinity_mode must be 'fused', 'split' or 'fp32' (got {affinity_mode!r})")
    b, n, d = feats.shape
    if upsample is not None:
        (hp, wp), (hl, wl) = upsample
        if hp * wp != n:
            raise ValueError(f"upsample grid {hp}x{wp} does not match N={n}")
        if normalize:
            feats = hip.normalize_rows(feats.contiguous())
        feats = torch.nn.functional.interpolate(feats.transpose(1, 2).reshape(b, d, hp, wp), size=(hl, wl),
                                                mode="bilinear", align_corners=False)
        feats = feats.reshape(b, d, hl * wl).transpose(1, 2).contiguous()
        normalize, n = False, hl * wl  # extract.py normalises BEFORE the resize and not again after it
    if not K < n:
        raise ValueError(f"need K < N (K={K}, N={n})")
    dense_all = K > MAX_LANCZOS_K and n > K + 2   # tiny N: the Krylov space is the whole space, the kernel handles it
    raw = problem.startswith("_raw_")  # internal: keep the solver's ranking order (used by the retry path)
    if raw:
        problem = problem[5:]
    if problem not in _PROBLEM_MODE:
        raise ValueError(f"unknown problem {problem!r}")
    if problem == "affinity_svd":
        threshold_at_zero = False  # the singular vectors of F are the eigenvectors of the UN-thresholded F F^T
    # extract.py:194 `W_feat = W_feat / W_feat.max()` (Laplacian branches only).  The kernels work on the unscaled W; the
    # division is put back on the outputs (reference_scale).  W.max() is max_i |f_i|^2 - Cauchy-Schwarz bounds every
    # entry by it and the diagonal attains it - which is 1 for normalised rows: only un-normalised or upsampled
    # (interpolated, not re-normalised) features need it.
    wmax = None
    if problem in ("laplacian", "laplacian_unnormalized") and not normalize and not raw:
        wmax = feats.square().sum(-1).amax(-1)
    # W as 16-bit fixed point (half the bytes of the solver's only HBM stream) whenever the problem allows it: the
    # normalised Laplacian is invariant to the scale of W, and normalised + thresholded similarities lie in [0, 1]
    # (after an upsample the rows are interpolated, not re-normalised: |w| <= 1 still holds, but keep f32 there).
    if w_dtype not in ("u16", "f32"):
        raise ValueError(f"w_dtype must be 'u16' or 'f32' (got {w_dtype!r})")
    w_u16 = (problem == "laplacian" and normalize and threshold_at_zero and upsample is None and d % 32 == 0
             and affinity_mode in ("fused", "split") and w_dtype == "u16")
    ld = hip.affinity_ld(n)
    per_image = hip.affinity_elems(n) * (2 if w_u16 else 4) + 2 * 66 * ld * 4
    chunk = max(1, min(b, max_bytes // per_image))
    evals, evecs, infos = [], [], []
    if dense_all:
        print(f"[dss] K={K} > {MAX_LANCZOS_K}: beyond the Krylov space of the Lanczos kernel - dense fp64 solve per image")
        ev, vec = dense_eigs(feats, K, normalize, threshold_at_zero, problem)
        evals.append(ev), evecs.append(vec), infos.append(torch.ones(b, dtype=torch.int32, device=feats.device))
    for s in range(0, 0 if dense_all else b, chunk):
        f = feats[s:s + chunk].contiguous()
        if affinity_mode == "fp32" or d % 32 != 0:  # exact fp32 MFMA (bitwise an fmaf chain), MFMA-bound
            if normalize:
                f = hip.normalize_rows(f)
            w = hip.affinity(f, threshold_at_zero)
        elif w_u16 and affinity_mode == "fused" and d >= 256:   # raw features -> packed 16-bit W in one kernel
            if feats16 is not None and rnorm is not None:
                w = hip.affinity_f16_u16(feats16[s:s + chunk].contiguous(), rnorm[s:s + chunk].contiguous())
            else:
                w = hip.affinity_fused_u16(f)
        else:  # split-f16 (fp32-class accuracy, ~1e-7), HBM-bound; fused with the row normalisation
            w = hip.affinity_split(f, normalize, threshold_at_zero, u16=w_u16)
        ev, vec, info = hip.laplacian_eigs(w, n, K, ncv=ncv, tol=tol, max_restarts=max_restarts,
                                           mode=_PROBLEM_MODE[problem])
        evals.append(ev), evecs.append(vec), infos.append(info)
        del w
    ev, vec, info = torch.cat(evals), torch.cat(evecs), torch.cat(infos)
    if retry and bool((info <= 0).any()):
        # The reference reacts to an ARPACK failure by re-solving in 'SM' mode (extract.py:228-229).  Here the
        # images that exhausted their restart budget are re-solved alone with the largest Krylov space and a
        # 10x restart budget; everything else keeps its first answer.
        bad = (info <= 0).nonzero().flatten()
        print(f"[dss] {bad.numel()} of {info.numel()} images exhausted the Lanczos restart budget "
              f"(passes {(-info[bad]).tolist()[:4]}...): re-solving them with ncv=64")
        ev2, vec2, info2 = laplacian_eigs_from_features(
            feats[bad], K, normalize=normalize, threshold_at_zero=threshold_at_zero, ncv=64,
            tol=tol, max_restarts=10 * (max_restarts if max_restarts > 0 else 60), max_bytes=max_bytes,
            strict=False, affinity_mode=affinity_mode, retry=False, problem="_raw_" + problem, upsample=None,
            w_dtype=w_dtype)
        ev[bad], vec[bad], info[bad] = ev2, vec2, info2
        still = (info <= 0).nonzero().flatten()
        if still.numel():   # last resort, like the reference's second eigsh call: it always produces an answer
            print(f"[dss] {still.numel()} image(s) still unconverged: dense fp64 solve for them")
            ev[still], vec[still] = dense_eigs(feats[still], K, normalize, threshold_at_zero, problem)
            info[still] = info[still].abs().clamp_min(1)
    if wmax is not None:
        ev, vec = reference_scale(problem, wmax, ev, vec)
    if not raw:
        ev, vec = _reference_order(problem, ev, vec)
    if strict:
        bad = (info <= 0).nonzero().flatten().tolist()
        if bad:
            raise EigsNotConverged(f"Lanczos did not converge for images {bad[:8]} (info={info[bad[:8]].tolist()})")
    return ev, vec, info


@torch.no_grad()
def feature_affinity_dense(feats: torch.Tensor, normalize: bool = True, threshold_at_zero: bool = True,
                           upsample: Optional[Tuple[Tuple[int, int], Tuple[int, int]]] = None) -> torch.Tensor:
    """extract.py:146-148,178-194 as a DENSE matrix on the device: ``W_feat = F F^T`` (rows normalised, resized to the
    low-resolution grid when ``upsample`` says so, thresholded at zero) divided by its maximum - ``[B, N, N]`` f32.
    Only the colour-fusion branch (``image_color_lambda > 0``) needs the matrix in this form; the default path keeps the
    packed tiles and never divides."""
    if feats.dim() == 2:
        feats = feats[None]
    b, n, d = feats.shape
    f = feats.contiguous()
    if normalize:
        f = hip.normalize_rows(f)
    if upsample is not None:
        (hp, wp), (hl, wl) = upsample
        f = torch.nn.functional.interpolate(f.transpose(1, 2).reshape(b, d, hp, wp), size=(hl, wl), mode="bilinear",
                                            align_corners=False)
        f = f.reshape(b, d, hl * wl).transpose(1, 2).contiguous()
        n = hl * wl
    wp_ = hip.affinity_split(f, False, threshold_at_zero) if d % 32 == 0 else hip.affinity(f, threshold_at_zero)
    w = hip.affinity_to_dense(wp_, n)[:, :n, :n]
    return w / w.amax(dim=(1, 2), keepdim=True)


@torch.no_grad()
def eigs_from_dense_affinity(w: torch.Tensor, K: int, problem: str = "laplacian", ncv: int = 0, tol: float = 0.0,
                             max_restarts: int = 0):
    """Eigenpairs of a general dense symmetric non-negative affinity ``[B, N, N]`` f32 (on the device) with the same
    Lanczos kernel as the hot path: ``(D - W) v = lambda D v`` (``problem="laplacian"``) or ``(D - W) v = lambda v``
    (``"laplacian_unnormalized"``), ``D = diag(row sums, < 1e-12 -> 1)``.  The body of extract.py:218-240 for a
    ``W_comb`` that is not a feature Gram matrix.  Returns ``(eigenvalues [B, K], eigenvectors [B, K, N], info [B])``;
    an image that exhausts its restart budget is re-solved with the largest Krylov space and then densely in fp64."""
    if problem not in ("laplacian", "laplacian_unnormalized"):
        raise ValueError(f"eigs_from_dense_affinity: unsupported problem {problem!r}")
    if w.dim() == 2:
        w = w[None]
    b, n, _ = w.shape
    if not K < n:
        raise ValueError(f"need K < N (K={K}, N={n})")
    info = torch.zeros(b, dtype=torch.int32, device=w.device)
    if K > MAX_LANCZOS_K and n > K + 2:
        ev = torch.empty((b, K), dtype=torch.float32, device=w.device)
        vec = torch.empty((b, K, n), dtype=torch.float32, device=w.device)
    else:
        wp = hip.affinity_from_dense(w)
        ev, vec, info = hip.laplacian_eigs(wp, n, K, ncv=ncv, tol=tol, max_restarts=max_restarts,
                                           mode=_PROBLEM_MODE[problem])
        bad = (info <= 0).nonzero().flatten()
        if bad.numel():
            print(f"[dss] {bad.numel()} of {b} images exhausted the Lanczos restart budget: re-solving with ncv=64")
            ev[bad], vec[bad], info[bad] = hip.laplacian_eigs(
                wp[bad].contiguous(), n, K, ncv=64, tol=tol, max_restarts=10 * (max_restarts if max_restarts > 0 else 60),
                mode=_PROBLEM_MODE[problem])
    for j in (info <= 0).nonzero().flatten().tolist():   # K > 62, or still unconverged: dense fp64, always an answer
        e, v = _dense_eigs_w(w[j].double(), K, problem)
        ev[j], vec[j] = e.float(), v.float()
        hip.sign_rule_(vec[j:j + 1])
        info[j] = max(1, abs(int(info[j])))
    return ev, vec, info


@torch.no_grad()
def single_region_masks(eigenvectors: torch.Tensor, threshold: float = 0.0) -> torch.Tensor:
    """extract.py:383-407 on the device, straight from the solver's output: u8 ``[B, N]`` masks (0 / 255) of
    ``eigenvectors[:, 1] > threshold`` - reshape ``(H_patch, W_patch)`` for the PNG the reference writes.  Host tensors (a
    machine without a GPU running the segmentation commands, as the reference did) take the same comparison in torch."""
    if not eigenvectors.is_cuda:
        return (eigenvectors[:, 1] > threshold).to(torch.uint8) * 255
    return hip.fiedler_mask(eigenvectors, 1, threshold)


def adaptive_num_segments(eigenvalues: torch.Tensor) -> List[int]:
    """``[B, K]`` ascending eigenvalues -> per image, one more than the position of the largest gap between consecutive
    eigenvalues, the gap behind eigenvalue 0 not counted (extract.py:306-309)."""
    gaps = torch.diff(eigenvalues.float(), dim=1)
    gaps[:, 0] = float("-inf")
    return (gaps.argmax(dim=1) + 1).tolist()


def border_owner_to_zero(labels: torch.Tensor) -> torch.Tensor:
    """``[rows, cols]`` integer labels -> the same map with the label that covers most of the image border exchanged with
    label 0 (the reference's background rule, extract.py:341-346 with extract_utils.py:124-135: the four border lines are
    concatenated, so corners count twice; ties go to the smaller label)."""
    border = torch.cat((labels[:, 0], labels[:, -1], labels[0, :], labels[-1, :])).long()
    owner = int(torch.bincount(border).argmax())
    out = labels.clone()
    out[labels == owner] = 0
    out[labels == 0] = owner
    return out


@torch.no_grad()
def kmeans_lloyd(points: torch.Tensor, k: int, seed: int = 0, max_iter: int = 300, tol: float = 1e-4, n_init: int = 1) -> torch.Tensor:
    """Plain Lloyd K-means on the device for what ``dss_kmeans_segments`` does not take (more than 8192 points / 64
    coordinates / 32 clusters; the reference's ``kmeans_baseline`` over raw 384-d features): k-means++ seeding from a
    seeded generator, iterations until the labels repeat or the squared centre shift is below ``tol`` x the mean
    coordinate variance (sklearn's two rules).  ``points [N, d]`` f32 -> ``[N]`` int64 labels.  ``n_init > 1``: that many runs
    from seeds ``seed, seed + 1, ..``, the tightest partition wins (what sklearn's ``n_init`` does; the host fallback of the
    segmentation commands uses 4)."""
    if n_init > 1:
        best, best_inertia = None, float("inf")
        for r in range(int(n_init)):
            lab = kmeans_lloyd(points, k, seed + r, max_iter, tol)
            cen = torch.zeros((int(lab.max()) + 1, points.shape[1]), dtype=points.dtype, device=points.device).index_add_(0, lab, points)
            cen = cen / torch.bincount(lab, minlength=cen.shape[0]).clamp(min=1).unsqueeze(1)
            inertia = float((points - cen[lab]).square().sum())
            if inertia < best_inertia:
                best, best_inertia = lab, inertia
        return best
    n, _ = points.shape
    k = max(1, min(int(k), n))
    gen = torch.Generator(device="cpu").manual_seed(int(seed))
    centres = points[torch.randint(n, (1,), generator=gen)]
    d2 = torch.cdist(points, centres).square().amin(dim=1)
    while centres.shape[0] < k:                                   # k-means++: next centre drawn proportionally to D^2
        total = float(d2.sum())
        pick = int(torch.searchsorted(torch.cumsum(d2, 0), torch.rand((), generator=gen).item() * total).clamp(max=n - 1)) \
            if total > 0 else int(torch.randint(n, (1,), generator=gen))
        centres = torch.cat((centres, points[pick:pick + 1]))
        d2 = torch.minimum(d2, (points - points[pick]).square().sum(dim=1))
    bar = tol * float(points.var(dim=0, unbiased=False).mean())
    labels = torch.full((n,), -1, dtype=torch.long, device=points.device)
    for _ in range(max_iter):
        new_labels = torch.cdist(points, centres).argmin(dim=1)
        sums = torch.zeros_like(centres).index_add_(0, new_labels, points)
        counts = torch.bincount(new_labels, minlength=k).unsqueeze(1)
        new_centres = torch.where(counts > 0, sums / counts.clamp(min=1), centres)     # an empty cluster keeps its centre
        shift = float((new_centres - centres).square().sum())
        done = bool((new_labels == labels).all()) or shift <= bar
        labels, centres = new_labels, new_centres
        if done:
            break
    return torch.cdist(points, centres).argmin(dim=1)


@torch.no_grad()
def multi_region_segments(eigenvalues: torch.Tensor, eigenvectors: torch.Tensor, grid: Tuple[int, int],
                          adaptive: bool = False, non_adaptive_num_segments: int = 4, infer_bg_index: bool = True,
                          num_eigenvectors: int = 1_000_000, init: Optional[torch.Tensor] = None, seed: int = 0):
    """extract.py:283-352 on the device: K-means over ``eigenvectors[b, 1:1+num_eigenvectors].T`` (``hip.kmeans_segments``:
    Lloyd + sklearn's stopping rules; ``init`` ``[B, k, dims]`` or k-means++ from ``seed``), the number of segments fixed or
    - ``adaptive`` - from the largest eigengap, then the border vote that renames the segment owning most of the border to
    0.  ``grid`` = (rows, cols) of the eigenvectors' patch grid.  Returns u8 labels ``[B, rows, cols]``.  Both the CLI
    command and ``extract_eigs --multi_region_dir`` come through here; images beyond the kernel's limits (8192 points, 64
    coordinates, 32 segments) take ``kmeans_lloyd`` + ``border_owner_to_zero``.  The reference's ``KMeans()`` is unseeded:
    partitions agree with it up to K-means' dependence on the initial centres."""
    b, k, n = eigenvectors.shape
    assert grid[0] * grid[1] == n, (grid, n)
    dims = min(int(num_eigenvectors), k - 1)
    out = torch.empty((b, n), dtype=torch.uint8, device=eigenvectors.device)
    ks = adaptive_num_segments(eigenvalues) if adaptive else [int(non_adaptive_num_segments)] * b
    for kk in sorted(set(ks)):
        idx = [i for i, v in enumerate(ks) if v == kk]
        sel = torch.tensor(idx, device=eigenvectors.device)
        if eigenvectors.is_cuda and n <= 8192 and 1 <= dims <= 64 and 1 <= kk <= 32:   # (host tensors: the tensor route below)
            lab, _, _ = hip.kmeans_segments(eigenvectors[sel].contiguous(), kk, first=1, dims=dims, grid=grid,
                                            infer_bg=infer_bg_index, init=None if init is None else init[sel], seed=seed)
            out[sel] = lab
        else:
            for i in idx:
                lab = kmeans_lloyd(eigenvectors[i, 1:1 + dims].t().contiguous(), kk, seed=seed,
                                   n_init=1 if eigenvectors.is_cuda else 4).view(grid)
                out[i] = (border_owner_to_zero(lab) if infer_bg_index else lab).reshape(-1).to(torch.uint8)
    return out.view(b, grid[0], grid[1])


def group_by_shape(shapes: List[Tuple[int, ...]], max_batch: int) -> List[List[int]]:
    """Indices grouped into batches of identical shape (order of first appearance preserved inside a
    batch; batches ordered by their first member)."""
    buckets, order = {}, []
    for i, s in enumerate(shapes):
        key = tuple(s)
        if key not in buckets or len(buckets[key][-1]) >= max_batch:
            buckets.setdefault(key, []).append([])
            order.append(buckets[key][-1])
        buckets[key][-1].append(i)
    return order
