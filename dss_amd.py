"""Import alias: the product package lives in ``deep-spectral-segmentation_amd/`` (a directory name
Python cannot import directly because of the hyphens).  ``import dss_amd`` loads that directory as
the package ``dss_amd`` (sub-modules: ``dss_amd.extract``, ``dss_amd.hip``, ...)."""
import importlib.util as _ilu
import pathlib as _pl
import sys as _sys

_pkg_dir = _pl.Path(__file__).resolve().parent / "deep-spectral-segmentation_amd"
_spec = _ilu.spec_from_file_location("dss_amd", _pkg_dir / "__init__.py",
                                     submodule_search_locations=[str(_pkg_dir)])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["dss_amd"] = _mod
_spec.loader.exec_module(_mod)
