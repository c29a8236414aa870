import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


    config.addinivalue_line("markers", "gpu_unvalidated: a GPU test that has not yet passed on hardware -- selected ONLY by `-m gpu_unvalidated` "
                            "(never by `-m gpu` or `-m 'not gpu'`); it becomes `gpu` once a gpurun session has passed it")


def pytest_collection_modifyitems(config, items):
    """Tests that have never met hardware stay out of the driver's gates (round-2 review: 28 such tests went straight into `-m gpu`)."""
    if "gpu_unvalidated" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpu_unvalidated") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def pytest_collection_finish(session):
    """On a fresh GPU box the very first `import torch` pages the image in and can take minutes: do it here, outside any test's
    timeout, when a selected test is going to need it (the product itself never imports torch)."""
    if os.environ.get("PLX_SKIP_TORCH_PREIMPORT") == "1":      # short targeted GPU sessions whose selected tests never touch torch
        return
    if any(item.get_closest_marker("gpu") or item.get_closest_marker("gpu_unvalidated") for item in session.items):
        try:
            import torch  # noqa: F401
        except Exception:
            pass


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def pl():
    """The product: polars_amd bound to GPU 0.  Fails loudly without the HIP library / a GPU."""
    import polars_amd
    polars_amd.init(0)
    return polars_amd
