"""CPU: the GPU eigensolver's source (csrc/eigs_core.h) compiled with g++ as a single-thread emulation and
checked against the reference goldens - validates the restart / Rayleigh-Ritz / sign-rule LOGIC of the
kernel without a GPU.  The emulation library is test infrastructure (tests/host_emul), never shipped."""
import ctypes
import glob
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests.util import check_eigs, golden_case, build_w64, d_orthonormality

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
                                            ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int]
    return lib


def pack_sym(w, ld):
    """Dense [n, n] -> packed upper-triangular 64x64 tiles (include/dss_hip.h, dss_affinity)."""
    n = w.shape[0]
    nt = ld // 64
    full = np.zeros((ld, ld), np.float32)
    full[:n, :n] = w
    tiles = [full[64 * i:64 * i + 64, 64 * j:64 * j + 64].reshape(-1) for i in range(nt) for j in range(i, nt)]
    return np.ascontiguousarray(np.concatenate(tiles))


def run_emul(lib, feats, K, ncv=0, keep=0, tol=2e-6, max_restarts=60):
    x = feats / np.maximum(np.linalg.norm(feats, axis=1, keepdims=True), 1e-12)
    x = x.astype(np.float32)
    w = x @ x.T
    w = w * (w > 0)
    n = w.shape[0]
    ld = (n + 63) // 64 * 64
    wp = pack_sym(w, ld)
    ncv = ncv or min(max(2 * K + 10, 20), 64, n)
    keep = keep or (ncv + K) // 2
    ev, vec, info = np.zeros(K, np.float32), np.zeros((K, n), np.float32), np.zeros(1, np.int32)
    lib.dss_emul_laplacian_eigs(wp.ctypes.data_as(FP), 1, n, ld, K, ev.ctypes.data_as(FP), vec.ctypes.data_as(FP),
                                info.ctypes.data_as(IP), ncv, keep, tol, max_restarts)
    return ev, vec, int(info[0])


CASES = [p for p in sorted(glob.glob(str(HERE / "golden" / "eigs_*.npz"))) if "3600" not in p]


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("eigs_")[-1][:-4])
def test_kernel_logic_matches_reference_goldens(emul, path):
    feats, K, ref_lam, ref_vec, _ = golden_case(path)
    lam, vec, info = run_emul(emul, feats, K)
    assert info > 0, f"not converged (info={info})"
    check_eigs(vec, lam, ref_vec, ref_lam, what=path)
    _, d = build_w64(feats)
    assert d_orthonormality(vec, d=d) < 1e-4
    for k in range(K):
        assert not (0.5 < np.mean(vec[k] > 0) < 1.0)


def test_kernel_logic_tiny_and_full_dimension(emul):
    """N smaller than the default Krylov dimension: the basis spans the whole space (breakdown path)."""
    rng = np.random.default_rng(0)
    feats = rng.normal(size=(16, 32)).astype(np.float32)
    from oracle.spectral_ref import dense_f64_eigs
    lam64, v64 = dense_f64_eigs(feats, 5)
    lam, vec, info = run_emul(emul, feats, 5)
    assert info > 0
    check_eigs(vec, lam, v64, lam64, what="tiny")


def test_kernel_logic_restart_budget_reports_nonconvergence(emul):
    feats, K, *_ = golden_case([p for p in CASES if "g2_random_900" in p][0])
    lam, vec, info = run_emul(emul, feats, K, max_restarts=1)
    assert info < 0 and np.all(np.isfinite(vec))
