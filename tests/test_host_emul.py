"""CPU: the GPU eigensolver's source (csrc/eigs_core.h) compiled with g++ as a single-thread emulation and
checked against the reference goldens - validates the restart / Rayleigh-Ritz / sign-rule LOGIC of the
kernel without a GPU.  The emulation library is test infrastructure (tests/host_emul), never shipped."""
import ctypes
import glob
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests.util import check_eigs, golden_case, golden_ext, build_w64, d_orthonormality

HERE = Path(__file__).resolve().parent
FP = ctypes.POINTER(ctypes.c_float)
IP = ctypes.POINTER(ctypes.c_int32)


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    out = tmp_path_factory.mktemp("emul") / "libeigs_emul.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(out),
                    str(HERE / "host_emul" / "eigs_emul.cpp")], check=True)
    lib = ctypes.CDLL(str(out))
    lib.dss_emul_laplacian_eigs.argtypes = [FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, FP, FP, IP,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]
    lib.dss_emul_laplacian_eigs_u16.argtypes = [ctypes.POINTER(ctypes.c_uint16), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, FP, FP, IP, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_int]
    return lib


def pack_sym(w, ld):
    """Dense [n, n] -> packed upper-triangular 64x64 tiles (include/dss_hip.h, dss_affinity)."""
    n = w.shape[0]
    nt = ld // 64
    full = np.zeros((ld, ld), np.float32)
    full[:n, :n] = w
    tiles = [full[64 * i:64 * i + 64, 64 * j:64 * j + 64].reshape(-1) for i in range(nt) for j in range(i, nt)]
    return np.ascontiguousarray(np.concatenate(tiles))


def run_emul(lib, feats, K, ncv=0, keep=0, tol=2e-6, max_restarts=60, mode=0, threshold=True, u16=False):
    x = feats / np.maximum(np.linalg.norm(feats, axis=1, keepdims=True), 1e-12)
    x = x.astype(np.float32)
    w = x @ x.T
    if threshold:
        w = w * (w > 0)
    n = w.shape[0]
    ld = (n + 63) // 64 * 64
    wp = pack_sym(w, ld)
    ncv = ncv or min(max(2 * K + 10, 20), 64, n)
    keep = keep or (ncv + K) // 2
    ev, vec, info = np.zeros(K, np.float32), np.zeros((K, n), np.float32), np.zeros(1, np.int32)
    if u16:  # the product path's storage: round(65535 w), w in [0, 1]
        assert threshold and mode == 0
        wq = np.ascontiguousarray(np.rint(np.clip(wp, 0.0, 1.0) * 65535.0).astype(np.uint16))
        lib.dss_emul_laplacian_eigs_u16(wq.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), 1, n, ld, K,
                                        ev.ctypes.data_as(FP), vec.ctypes.data_as(FP), info.ctypes.data_as(IP), ncv,
                                        keep, tol, max_restarts)
        return ev, vec, int(info[0])
    lib.dss_emul_laplacian_eigs(wp.ctypes.data_as(FP), 1, n, ld, K, ev.ctypes.data_as(FP), vec.ctypes.data_as(FP),
                                info.ctypes.data_as(IP), ncv, keep, tol, max_restarts, mode)
    return ev, vec, int(info[0])


CASES = [p for p in sorted(glob.glob(str(HERE / "golden" / "eigs_*.npz"))) if "3600" not in p]


@pytest.mark.parametrize("u16", [False, True], ids=["w_f32", "w_u16"])
@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("eigs_")[-1][:-4])
def test_kernel_logic_matches_reference_goldens(emul, path, u16):
    """u16: W quantised to 16-bit fixed point as the product path stores it - same tolerance against the reference's
    own outputs (the quantisation moves the eigenvectors by <= 1e-6 in cosine), same D-orthonormality in TRUE units."""
    feats, K, ref_lam, ref_vec, g = golden_case(path)
    lam, vec, info = run_emul(emul, feats, K, u16=u16)
    assert info > 0, f"not converged (info={info})"
    _, d = build_w64(feats)
    check_eigs(vec, lam, ref_vec, ref_lam, what=path, d=d, ext=golden_ext(g))
    assert d_orthonormality(vec, d=d) < 1e-4
    for k in range(K):
        assert not (0.5 < np.mean(vec[k] > 0) < 1.0)


def test_kernel_logic_tiny_and_full_dimension(emul):
    """N smaller than the default Krylov dimension: the basis spans the whole space (breakdown path)."""
    rng = np.random.default_rng(0)
    feats = rng.normal(size=(16, 32)).astype(np.float32)
    from oracle.spectral_ref import dense_f64_eigs
    lam64, v64 = dense_f64_eigs(feats, 8)
    lam, vec, info = run_emul(emul, feats, 5)
    assert info > 0
    check_eigs(vec, lam, v64[:5], lam64[:5], what="tiny", ext=(lam64, v64))


def test_kernel_logic_restart_budget_reports_nonconvergence(emul):
    feats, K, *_ = golden_case([p for p in CASES if "g2_random_900" in p][0])
    lam, vec, info = run_emul(emul, feats, K, max_restarts=1)
    assert info < 0 and np.all(np.isfinite(vec))


# ---- the other _extract_eig branches (extract.py:159-172, :230-234): same kernel, other operator / selection modes ----
# (upsampled / un-normalised features: the resize and the W.max() rescaling are host-side steps of spectral.py,
#  covered on the GPU by tests/test_gpu_kernels.py::test_other_branches_match_reference_goldens)
MODE_FILES = [p for p in sorted(glob.glob(str(HERE / "golden" / "modes_*.npz")))
              if "upsample" not in p and "nonorm" not in p]


def mode_outputs_like_reference(kind, values, vectors):
    """Arrange (values, vectors) as the reference saves them for this branch (values/vectors arrive in the kernel's
    ranking order: descending theta, or descending |theta| for the affinity modes)."""
    values, vectors = np.asarray(values, np.float64), np.asarray(vectors)
    if kind == "affinity":          # eigsh ascending values kept as-is, eigenvectors FLIPPED to descending (:171-172)
        order = np.argsort(values)
        return values[order], vectors[order][::-1]
    if kind == "affinity_svd":      # singular values descending, left singular vectors (:161-163)
        order = np.argsort(-values)
        return np.sqrt(np.maximum(values[order], 0.0)), vectors[order]
    return values, vectors          # lapnorm=False: eigenvalues ascending (:232-235)


def check_mode_against_golden(g, values, vectors, what):
    kw = eval(str(g["kwargs"]))     # fixture metadata written by oracle/make_golden.py
    kind = kw["which_matrix"] if kw["which_matrix"] != "laplacian" else "laplacian_unnormalized"
    lam, vec = mode_outputs_like_reference(kind, values, vectors)
    ref_lam, ref_vec = np.asarray(g["eigenvalues"], np.float64), g["eigenvectors"]
    scale = max(1.0, np.abs(ref_lam).max())
    if kind == "affinity":          # vectors are in the opposite order of the values: compare in a common order
        check_eigs(vec[::-1], lam, ref_vec[::-1], ref_lam, what=what, lam_tol=2e-5 * scale, gap_tol=1e-4 * scale)
    elif kind == "affinity_svd":
        check_eigs(vec[::-1], lam[::-1], ref_vec[::-1], ref_lam[::-1], what=what, lam_tol=2e-5 * scale,
                   gap_tol=1e-4 * scale)
    else:
        check_eigs(vec, lam, ref_vec, ref_lam, what=what, lam_tol=2e-5 * scale, gap_tol=1e-4 * scale)


@pytest.mark.parametrize("path", MODE_FILES, ids=lambda p: p.split("modes_")[-1][:-4])
def test_kernel_logic_other_branches_match_reference_goldens(emul, path):
    g = np.load(path)
    feats = __import__("dss_amd").synthetic.synthetic_features(str(g["kind"]), int(g["n"]), int(g["d"]), int(g["seed"]),
                                                              tuple(g["hw"]))
    kw = eval(str(g["kwargs"]))
    K = int(g["K"])
    if kw["which_matrix"] == "affinity":
        mode, thr = 1, kw.get("threshold_at_zero", True)
    elif kw["which_matrix"] == "affinity_svd":
        mode, thr = 1, False
    else:
        mode, thr = 2, True
    values, vectors, info = run_emul(emul, feats, K, mode=mode, threshold=thr)
    assert info > 0, info
    check_mode_against_golden(g, values, vectors, path)


@pytest.mark.parametrize("n,K,ncv", [(15, 3, 0), (33, 4, 13), (70, 6, 21), (200, 5, 17), (130, 12, 33)])
def test_kernel_logic_odd_krylov_dimensions(emul, n, K, ncv):
    """Odd projected dimensions exercise the dummy index of the Jacobi pairing (a pair whose partner does not exist is
    the identity; 2x2 block updates skip its row and column), with and without restarts (small ncv) and in the breakdown
    case N = ncv.  Against dense fp64 on the same features."""
    from oracle.spectral_ref import dense_f64_eigs

    rng = np.random.default_rng(100 + n)
    base = rng.normal(size=(n, 24)).astype(np.float32)
    feats = base + 0.6 * rng.normal(size=(1, 24)).astype(np.float32)    # a common component: mostly positive affinities
    k2 = min(n - 1, K + 6)
    lam64, v64 = dense_f64_eigs(feats, k2)
    lam, vec, info = run_emul(emul, feats, K, ncv=ncv)
    assert info > 0, info
    check_eigs(vec, lam, v64[:K], lam64[:K], what=f"n{n}_K{K}_ncv{ncv}", ext=(lam64, v64))
