import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# BOXTREE_EMU=1: run the tests -- the `-m gpu` ones included -- against tests/emu's CPU emulation of the
# kernels instead of a GPU (test infrastructure for boxes without one: the kernels' logic against
# the oracle, nothing about their speed; tests/emu/README.md)
EMU = os.environ.get("BOXTREE_EMU", "0") == "1"
if EMU:
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import emu_actx
    emu_actx.install_for_tests()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # marks the reference's test files carry (tests/test_reference_suite.py compiles them)
    for name in ("opencl", "area_query", "geo_lookup", "mpi"):
        config.addinivalue_line("markers", f"{name}: mark of the reference's own tests")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); compiled on first use with gcc."""
    from oracle import oracle as orc
    orc.build_lib()
    return orc
