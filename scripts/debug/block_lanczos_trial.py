"""CPU trial of the BLOCK variant of the eigensolver: how many passes over W would a block Lanczos with
b = 2 / 4 / 8 vectors per pass need for the K wanted pairs, against the passes the single-vector thick-restart solver of
csrc/eigs_core.h takes (its host emulation, tests/host_emul) on the same matrices?

A pass over W is what the GPU kernel pays for (0.87 MB per image at N = 900 as 16-bit W); a block step reads W ONCE for b
vectors (through v_mfma_f32_32x32x2_f32 on the GPU), so passes = block steps.  The block solver here is the most favourable
one: fp64, full reorthogonalisation, a Rayleigh-Ritz check after EVERY block step with the kernel's own criterion
(|A y - theta y| <= tol max(|theta|, 1e-3 theta_max), tol = 2e-6), the basis allowed to grow to the kernel's 64 vectors and one
thick restart (keep the K + b best Ritz vectors) when it is full.

    python scripts/debug/block_lanczos_trial.py > profiles/r05_block_lanczos_trial.txt
"""
import ctypes, glob, os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import golden_case  # noqa: E402
from tests import test_host_emul as the  # noqa: E402
import dss_amd  # noqa: E402,F401
from dss_amd import synthetic  # noqa: E402


def operator(feats):
    x = feats / np.maximum(np.linalg.norm(feats, axis=1, keepdims=True), 1e-12)
    w = (x.astype(np.float32) @ x.astype(np.float32).T).astype(np.float64)
    w = w * (w > 0)
    dis = 1.0 / np.sqrt(np.maximum(w.sum(1), 1e-12))
    return dis[:, None] * w * dis[None, :]                # D^-1/2 W D^-1/2: its LARGEST pairs are the Laplacian's smallest


def block_lanczos_passes(a, k, b, tol=2e-6, max_basis=64, seed=0):
    n = a.shape[0]
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.uniform(-1, 1, (n, b)))
    basis, passes = q, 0
    aq_all = np.zeros((n, 0))
    while passes < 200:
        aq = a @ basis[:, -b:] if aq_all.shape[1] else a @ basis       # ONE pass over W for b vectors
        passes += 1
        aq_all = np.concatenate([aq_all, aq], 1)
        t = basis.T @ aq_all
        t = 0.5 * (t + t.T)
        th, s = np.linalg.eigh(t)
        order = np.argsort(-th)[:k]
        y, ay = basis @ s[:, order], aq_all @ s[:, order]
        res = np.linalg.norm(ay - y * th[order], axis=0)
        bar = tol * np.maximum(np.abs(th[order]), 1e-3 * np.abs(th).max())
        if basis.shape[1] >= k and np.all(res <= bar):
            return passes
        w = aq - basis @ (basis.T @ aq)
        w -= basis @ (basis.T @ w)
        qn, r = np.linalg.qr(w)
        if basis.shape[1] + b > max_basis:                               # thick restart: keep the K + b best Ritz vectors
            keep = np.argsort(-th)[:k + b]
            basis, aq_all = basis @ s[:, keep], aq_all @ s[:, keep]
            w = qn - basis @ (basis.T @ qn)
            qn, _ = np.linalg.qr(w)
        basis = np.concatenate([basis, qn], 1)
    return -passes


def main():
    out = tempfile.mkdtemp()
    lib = os.path.join(out, "libeigs_emul.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", lib,
                    os.path.join(ROOT, "tests", "host_emul", "eigs_emul.cpp")], check=True)
    emul = ctypes.CDLL(lib)
    FP, IP = the.FP, the.IP
    emul.dss_emul_laplacian_eigs.argtypes = [FP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, FP, FP, IP,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]
    emul.dss_emul_laplacian_eigs_u16.argtypes = [ctypes.POINTER(ctypes.c_uint16), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, FP, FP, IP, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                 ctypes.c_int]
    cases = []
    for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "eigs_*.npz"))):
        if "3600" in p or "1600" in p:
            continue
        feats, k, *_ = golden_case(p)
        cases.append((os.path.basename(p)[5:-4], feats, k))
    for seed in range(6):      # the bench's kind of input: 30 x 30 patch grids, D = 384
        cases.append((f"blobs_900_seed{seed}", synthetic.synthetic_features("blobs", 900, 384, 700 + seed, (30, 30)), 5))
    print(f"{'matrix':28s} {'N':>5s} {'K':>3s} | single-vector thick restart (kernel logic, host emulation): W passes | block b=2  b=4  b=8 (W passes = block steps; x b matvecs)")
    tot = {"single": 0, 2: 0, 4: 0, 8: 0}
    for name, feats, k in cases:
        _, _, info = the.run_emul(emul, feats, k, u16=True)
        a = operator(feats)
        bl = {b: block_lanczos_passes(a, k, b) + 1 for b in (2, 4, 8)}      # + 1: the degree pass (W 1), as the kernel counts it
        tot["single"] += abs(info)
        for b in bl:
            tot[b] += bl[b]
        print(f"{name:28s} {feats.shape[0]:5d} {k:3d} | {info:6d} | " + "  ".join(f"{bl[b]:4d} ({(bl[b] - 1) * b:3d} mv)" for b in (2, 4, 8)))
    n = len(cases)
    print(f"{'mean':28s} {'':5s} {'':3s} | {tot['single'] / n:6.1f} | " + "  ".join(f"{tot[b] / n:4.1f}" + " " * 9 for b in (2, 4, 8)))


if __name__ == "__main__":
    main()
