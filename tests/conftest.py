import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """The library is a build product (git-ignored): compile it once if this checkout does not have it yet, so that
    the suites do not depend on test order.  A failed build is left to the tests that need the library to report."""
    lib = os.path.join(ROOT, "dl_ofdm_amd", "lib", "libdccn.so")
    if not os.path.exists(lib):
        try:
            import __graft_entry__ as g
            g.build()
        except Exception as e:                                   # noqa: BLE001
            print("conftest: building libdccn.so failed: %r" % (e,))


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a visible device: on a box without one they are skipped, not failed."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:                                            # noqa: BLE001
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible (run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
