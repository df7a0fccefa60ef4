"""MI355X-native implementation of the deep-spectral-segmentation ``extract.py`` hot path
(``extract_features`` -> ``extract_eigs``).  See DESIGN.md.

Sub-modules are imported lazily by the callers; importing this package never touches the GPU
and never needs the HIP library (calling an op without it raises ``HipLibraryError``)."""
import os as _os

__all__ = ["synthetic"]

# hipBLASLt's gfx950 kernels are all Stream-K-capable: they split the last, partly filled round of output tiles across workgroups,
# and that exchange is not reproducible on this stack (one launch in ~56 000 returns different values in whole tiles:
# profiles/r06_forward_stress.txt).  Tensile's switch below makes every launch data-parallel; it has to be in the environment
# before hipBLASLt serves its first GEMM in the process, so it is set when the package is imported (no overwrite: an explicit
# value of the caller's stands).  `dss_linear_lt` (csrc/gemm.hip) sets it too and VERIFIES it per problem: it only takes a
# candidate for which the library reports no partial-tile workspace, and fails loudly otherwise.
_os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")


def __getattr__(name):   # `dss_amd.synthetic` without importing torch at package import (pthfast's loader processes)
    if name == "synthetic":
        import importlib
        return importlib.import_module(".synthetic", __name__)
    raise AttributeError(name)
