"""Shared helpers for the parity tests (TEST INFRASTRUCTURE)."""
from __future__ import annotations

import numpy as np

import dss_amd  # noqa: F401
from dss_amd import synthetic
from oracle.spectral_ref import cos_err

COS_TOL = 1e-4   # BASELINE.json: eigenvectors within 1e-4 cosine of the reference CPU path
LAM_TOL = 1e-5   # SURVEY.md §8c comparison rule for eigenvalues
GAP_TOL = 1e-4   # below this eigenvalue gap individual eigenvectors are ill-conditioned:
                 # the reference's own fp32 ARPACK output is then > 1e-4 away from the fp64 truth


def golden_case(path):
    g = np.load(path)
    feats = synthetic.synthetic_features(str(g["kind"]), int(g["n"]), int(g["d"]), int(g["seed"]), tuple(g["hw"]))
    return feats, int(g["K"]), g["eigenvalues"], g["eigenvectors"], g


def check_eigs(vec, lam, ref_vec, ref_lam, what="", cos_tol=COS_TOL, lam_tol=LAM_TOL, gap_tol=GAP_TOL):
    """Assert parity of ``[K, N]`` eigenvectors / ``[K]`` eigenvalues with a reference set.

    Per vector ``1 - |cos| <= cos_tol``.  A vector that fails individually must belong to a cluster of
    eigenvalues chained by gaps < ``gap_tol``; then it must lie (to ``cos_tol``) in the span of the
    reference vectors of that cluster widened by one neighbour on each side - the principal-angle rule of
    SURVEY.md §8c.  Returns the per-vector cosine errors."""
    vec, ref_vec = np.asarray(vec, np.float64), np.asarray(ref_vec, np.float64)
    lam, ref_lam = np.asarray(lam, np.float64), np.asarray(ref_lam, np.float64)
    K = vec.shape[0]
    assert vec.shape == ref_vec.shape, (vec.shape, ref_vec.shape)
    assert np.all(np.isfinite(vec)) and np.all(np.isfinite(lam)), f"{what}: non-finite output"
    dl = np.abs(lam - ref_lam)
    assert dl.max() <= lam_tol, f"{what}: eigenvalue mismatch {dl.max():.2e} > {lam_tol} ({lam} vs {ref_lam})"
    assert np.all(np.diff(lam) >= -1e-6), f"{what}: eigenvalues not ascending: {lam}"
    ce = cos_err(vec, ref_vec)
    gaps = np.abs(np.diff(ref_lam))
    for i in np.nonzero(ce > cos_tol)[0]:
        lo = hi = i
        while lo > 0 and gaps[lo - 1] < gap_tol:
            lo -= 1
        while hi < K - 1 and gaps[hi] < gap_tol:
            hi += 1
        assert hi > lo, f"{what}: vector {i} cos_err {ce[i]:.2e} > {cos_tol} and its eigenvalue is isolated " \
                        f"(gaps {gaps[max(i - 1, 0):i + 1]})"
        lo2, hi2 = max(lo - 1, 0), min(hi + 1, K - 1)
        basis = ref_vec[lo2:hi2 + 1].T
        coef, *_ = np.linalg.lstsq(basis, vec[i], rcond=None)
        proj = basis @ coef
        err = 1.0 - np.linalg.norm(proj) / np.linalg.norm(vec[i])
        touches_edge = hi == K - 1  # the cluster may continue past K: its last members can mix with unseen ones
        assert err <= cos_tol or touches_edge, \
            f"{what}: vector {i} (cluster {lo}..{hi}, gaps<{gap_tol}) is {err:.2e} outside the reference span"
    return ce


def d_orthonormality(vec, feats_norm_w=None, d=None):
    vec = np.asarray(vec, np.float64)
    g = (vec * d[None, :]) @ vec.T
    return np.abs(g - np.eye(vec.shape[0])).max()


def build_w64(feats):
    x = np.asarray(feats, np.float64)
    x = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    w = x @ x.T
    w = w * (w > 0)
    d = w.sum(1)
    d[d < 1e-12] = 1.0
    return w, d
