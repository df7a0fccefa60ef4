"""Build libdss_hip.so (all HIP kernels + the C ABI of include/dss_hip.h) for gfx950 with hipcc.

    python deep-spectral-segmentation_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The library is written in-tree (``lib/libdss_hip.so``) so that it
travels to the GPU box with the repository snapshot; it is git-ignored (history stays source-only)."""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libdss_hip.so"
SOURCES = ["lib.hip", "preprocess.hip", "layernorm.hip", "attention.hip", "linear384.hip", "affinity.hip", "eigs.hip", "segment.hip", "gemm.hip"]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# attention.hip: the softmax max-chains read MFMA results; in IEEE mode hipcc quiets every such value with an extra
# v_max (64 of ~400 VALU instructions per key tile in a VALU-bound kernel).  No NaN can occur there (masking uses
# -inf), so NaN-honouring and IEEE mode are switched off for this file only.
EXTRA_FLAGS = {"attention.hip": ["-fno-honor-nans", "-mno-amdgpu-ieee"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [PKG.parent / "include" / "dss_hip.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    LIBDIR.mkdir(exist_ok=True)
    stamp = LIBDIR / "libdss_hip.sha256"
    digest = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text().strip() == digest:
        if verbose:
            print(f"[build] {LIB} is up to date")
        return LIB
    hipcc = _hipcc()
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)

    def compile_one(src: str) -> Path:
        obj = objdir / (Path(src).stem + ".o")
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", str(CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # gemm.hip calls hipBLASLt (the Linear layers that are not hand-written kernels, with an algorithm chosen here); under PyTorch
    # the soname resolves to the copy torch has already loaded, stand-alone to /opt/rocm's
    rocm_lib = str(Path(hipcc).resolve().parent.parent / "lib")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", str(LIB), *map(str, objs), "-L" + rocm_lib, "-lhipblaslt",
           "-Wl,-rpath," + rocm_lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(digest + "\n")
    if verbose:
        print(f"[build] wrote {LIB} ({LIB.stat().st_size} bytes)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
