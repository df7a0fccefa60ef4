"""MI355X-native implementation of the deep-spectral-segmentation ``extract.py`` hot path
(``extract_features`` -> ``extract_eigs``).  See DESIGN.md.

Sub-modules are imported lazily by the callers; importing this package never touches the GPU
and never needs the HIP library (calling an op without it raises ``HipLibraryError``)."""
from . import synthetic  # noqa: F401

__all__ = ["synthetic"]
