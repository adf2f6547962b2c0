import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The suite runs in several worker processes (pytest-xdist, `addopts` of pytest.ini: the parity cases spend their time in the CPU oracle, one
# case per worker instead of one after the other).  A worker's OpenMP teams (oracle, oracle/_ref) get their share of the host cores, not all
# of them: eight teams of 256 threads on 256 cores spend their time in barriers.
# (Measured the hard way, session r05_a: eight workers x 64 threads on 256 cores — twice the cores — ran the oracle FIFTEEN times slower, its
# teams spinning in each other's barriers; the sum over the workers must stay within the cores, and waiting threads must sleep.)
_workers = int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "0") or 0)
if _workers > 1 and "OMP_NUM_THREADS" not in os.environ:
    os.environ["OMP_NUM_THREADS"] = str(max(2, (os.cpu_count() or 8) // _workers))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("GOMP_SPINCOUNT", "1000")


def pytest_xdist_auto_num_workers(config):
    """`-n auto` (pytest.ini): eight workers on the GPU box's 256 host cores, a quarter of the cores on small machines"""
    n = os.cpu_count() or 8
    return 8 if n >= 32 else max(2, n // 4)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not os.environ.get("PYTEST_XDIST_WORKER"):
        # the controlling process runs no test itself: load the product library here too, so that "which native code did pytest load" has the
        # same answer whether or not the tests were handed to workers (a missing library is the tests' business: they fail loudly)
        try:
            from alicevision_amd import abi
            abi.load()
        except Exception:
            pass


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle.load()
