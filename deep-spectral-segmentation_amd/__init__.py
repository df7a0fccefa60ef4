"""MI355X-native implementation of the deep-spectral-segmentation ``extract.py`` hot path
(``extract_features`` -> ``extract_eigs``).  See DESIGN.md.

Sub-modules are imported lazily by the callers; importing this package never touches the GPU
and never needs the HIP library (calling an op without it raises ``HipLibraryError``)."""
__all__ = ["synthetic"]


def __getattr__(name):   # `dss_amd.synthetic` without importing torch at package import (pthfast's loader processes)
    if name == "synthetic":
        import importlib
        return importlib.import_module(".synthetic", __name__)
    raise AttributeError(name)
