"""Shared helpers for the parity tests (TEST INFRASTRUCTURE)."""
from __future__ import annotations

import numpy as np

import dss_amd  # noqa: F401
from dss_amd import synthetic
from oracle.spectral_ref import cos_err, edge_window_end, eig_clusters, subspace_err

COS_TOL = 1e-4   # BASELINE.json: eigenvectors within 1e-4 cosine of the reference CPU path
LAM_TOL = 1e-5   # SURVEY.md §8c comparison rule for eigenvalues
GAP_TOL = 1e-4   # below this eigenvalue gap individual eigenvectors are ill-conditioned:
                 # the reference's own fp32 ARPACK output is then > 1e-4 away from the fp64 truth


ORACLE_DRAWS = []   # draws of every oracle_target() call of the running test (tests/conftest.py reads and clears it)


def oracle_target(feats, K, **kw):
    """``oracle.spectral_ref.ref_laplacian_eigs_ext`` with book-keeping: which target an image was judged against - the
    reference's own ARPACK output (``draws`` >= 1: the draw that was within 1e-5 of fp64 on the isolated vectors) or the fp64
    dense solution substituted for it when every draw was bad (``draws`` < 0).  The per-test tally is printed, attached to
    the report and bounded by tests/conftest.py (``@pytest.mark.oracle_substitute(max_share=...)``, default 1/3)."""
    from oracle import spectral_ref

    lam, vec, ext, draws = spectral_ref.ref_laplacian_eigs_ext(feats, K, **kw)
    ORACLE_DRAWS.append(int(draws))
    return lam, vec, ext, draws


def golden_case(path):
    """Features + expected outputs of one golden file: ``(feats, K, eigenvalues [K], eigenvectors [K, N], npz)``.
    ``golden_ext(npz)`` gives the K + E pairs the reference produced when asked for K + E."""
    g = np.load(path)
    feats = synthetic.synthetic_features(str(g["kind"]), int(g["n"]), int(g["d"]), int(g["seed"]), tuple(g["hw"]))
    return feats, int(g["K"]), g["eigenvalues"], g["eigenvectors"], g


def golden_ext(g):
    """``(eigenvalues [K + E], eigenvectors [K + E, N])`` of the reference asked for K + E pairs (oracle/make_golden.py),
    or None for a golden without them: the extra pairs let ``check_eigs`` decide a cluster of near-equal eigenvalues
    that straddles index K - 1 instead of waving it through."""
    return (g["eigenvalues_ext"], g["eigenvectors_ext"]) if "eigenvectors_ext" in g else None


def check_eigs(vec, lam, ref_vec, ref_lam, what="", cos_tol=COS_TOL, lam_tol=LAM_TOL, gap_tol=GAP_TOL, d=None,
               ext=None, report=None):
    """Assert parity of ``[K, N]`` eigenvectors / ``[K]`` eigenvalues with the reference's ``[K, N]`` / ``[K]``.
    ``ext = (eigenvalues [K + E], eigenvectors [K + E, N])`` is the SAME reference asked for E more pairs (its first K
    agree with ``ref_*`` to the reference's own run-to-run noise, ~1e-5 in the eigenvalues).  EVERY branch carries a bound:

    * eigenvalues: ``|lam - ref_lam| <= lam_tol`` and ascending;
    * the reference eigenvalues (``ext`` when given) are split into clusters chained by gaps ``< gap_tol``;
    * an isolated eigenvalue: ``1 - |cos| <= cos_tol`` for its vector (the BASELINE.json metric);
    * a cluster inside ``[0, K)``: the largest principal angle between the two spans, measured in the D inner product
      when the degrees ``d`` are given (the eigenvectors are D-orthonormal), ``1 - cos(theta_max) <= cos_tol``;
    * a cluster that straddles K - 1: the computed members must lie, to the same bound, inside the span of the
      reference vectors from the start of the cluster up to the last one whose eigenvalue is within ``gap_tol`` of
      eigenvalue K - 1 (what a perturbation eps of the matrix leaks into the vectors beyond that window is bounded
      by eps / gap_tol, the conditioning the isolated case is held to).  The reference therefore has to reach
      ``lam[K - 1] + gap_tol``; if it does not - no ``ext``, or too few extra pairs - the check FAILS (ask the oracle
      for more: ``ref_laplacian_eigs_ext``).

    Returns the per-vector cosine errors; ``report`` (a list) receives one dict per cluster."""
    vec, ref_vec = np.asarray(vec, np.float64), np.asarray(ref_vec, np.float64)
    lam, ref_lam = np.asarray(lam, np.float64), np.asarray(ref_lam, np.float64)
    K = vec.shape[0]
    assert vec.shape == ref_vec.shape and ref_lam.shape[0] == K, (vec.shape, ref_vec.shape)
    x_lam, x_vec = (ref_lam, ref_vec) if ext is None else (np.asarray(ext[0], np.float64), np.asarray(ext[1], np.float64))
    kx = x_vec.shape[0]
    assert kx >= K and x_vec.shape[1] == vec.shape[1] and x_lam.shape[0] == kx, (vec.shape, x_vec.shape)
    assert np.all(np.isfinite(vec)) and np.all(np.isfinite(lam)), f"{what}: non-finite output"
    dl = np.abs(lam - ref_lam)
    assert dl.max() <= lam_tol, f"{what}: eigenvalue mismatch {dl.max():.2e} > {lam_tol} ({lam} vs {ref_lam})"
    assert np.all(np.diff(lam) >= -1e-6), f"{what}: eigenvalues not ascending: {lam}"
    ce = cos_err(vec, ref_vec)
    for lo, hi in eig_clusters(x_lam, gap_tol):
        if lo >= K:
            break
        if lo == hi:
            err, kind = float(ce[lo]), "isolated"
        elif hi < K:
            err, kind = subspace_err(vec[lo:hi + 1], ref_vec[lo:hi + 1], d), "cluster"
        else:
            win = edge_window_end(x_lam, K, gap_tol)
            assert win < kx - 1 or kx == vec.shape[1], \
                f"{what}: eigenvalue cluster {lo}..{hi} (gaps < {gap_tol}) straddles K - 1 = {K - 1} and the {kx} " \
                f"reference pairs end inside the {gap_tol} window above eigenvalue {K - 1}: the reference must be " \
                f"computed with more extra pairs to decide vectors {lo}..{K - 1}"
            hi = min(hi, win)
            err, kind = subspace_err(vec[lo:K], x_vec[lo:hi + 1], d), "edge-cluster"
        if report is not None:
            report.append({"first": int(lo), "last": int(hi), "kind": kind, "err": float(err),
                           "max_vector_cos_err": float(ce[lo:min(hi, K - 1) + 1].max())})
        assert err <= cos_tol, \
            f"{what}: {kind} {lo}..{hi}: error {err:.2e} > {cos_tol} (per-vector cos errors {ce[lo:min(hi, K - 1) + 1]}, " \
            f"eigenvalues {x_lam[lo:hi + 1]})"
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


def gelu_f16_poly(x16):
    """csrc/kres.h `gelu_poly_f16xn` in numpy, operation for operation (every f16 operation rounds once: products and sums are
    formed in float64, where they are exact, and rounded to f16): GELU on packed f16 as the Linear kernel's `gelu = 2` epilogue
    computes it from the f16-rounded pre-activation ``x16``."""
    import numpy as np

    f16 = np.float16
    bits = lambda b: np.array(b, dtype=np.uint16).view(f16)[()]
    x = np.asarray(x16, dtype=f16)
    x64 = x.astype(np.float64)
    a = np.minimum(np.abs(x64), 4.25)
    t = (a * float(bits(0x3388))).astype(f16).astype(np.float64)
    q = (float(bits(0x3f09)) * t + float(bits(0xc3b6))).astype(f16).astype(np.float64)
    for c in (0x3c3f, 0x4167, 0xbcb7, 0xbcc0, 0x39a8):
        q = (q * t + float(bits(c))).astype(f16).astype(np.float64)
    q = (q * q).astype(f16).astype(np.float64)
    return (-a * q + np.maximum(x64, 0.0)).astype(f16)
