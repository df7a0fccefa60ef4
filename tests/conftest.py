import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("DSS_ASSUME_YES", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "oracle_substitute(max_share): largest share of a test's images that may be judged "
                                       "against the fp64 substitute instead of the reference's ARPACK output (default 1/3)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return REPO / "tests" / "golden"


@pytest.fixture(autouse=True)
def _oracle_target_tally(request, record_property):
    """Every test that compares with the oracle through tests.util.oracle_target says, and bounds, how many of its images
    were judged against the reference's own ARPACK output and how many against the fp64 dense solution that replaces a
    reference whose every draw was bad (oracle/spectral_ref.ref_laplacian_eigs_ext, `draws` < 0)."""
    from tests import util

    util.ORACLE_DRAWS.clear()
    yield
    draws = list(util.ORACLE_DRAWS)
    util.ORACLE_DRAWS.clear()
    if not draws:
        return
    sub = sum(1 for d in draws if d < 0)
    redrawn = sum(1 for d in draws if d > 1)
    line = (f"[oracle targets] {request.node.name}: {len(draws) - sub} image(s) vs the reference's ARPACK output "
            f"({redrawn} after a re-draw), {sub} vs the fp64 substitute")
    print(line)
    record_property("oracle_targets", line)
    if sub:
        import warnings

        warnings.warn(line)
    mark = request.node.get_closest_marker("oracle_substitute")
    share = mark.kwargs.get("max_share", 1 / 3) if mark else 1 / 3
    assert sub <= share * len(draws) + 1e-9, f"{line}: more than {share:.2f} of the images were not judged against the reference itself"
