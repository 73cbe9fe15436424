"""pytest configuration: registers the `gpu` marker and the shared fixtures (golden vectors, oracle access)."""

import os
import sys

import numpy as np
import pytest

# The CPU oracle mirrors the reference's `omp atomic` updates; on a 128-core GPU box they contend so badly that small problems
# run slower than on 8 cores.  Cap the oracle's threads for the tests (bench.py's cpu_baseline leg is a separate process setting).
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
# idle OpenMP workers must sleep, not spin: once a test has imported torch its runtime's spinning threads starve the oracle's
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

# parameter sets of the golden vectors (tests/golden/make_golden.py)
PARAM_SETS = {
    "ref": dict(degree=2, gamma=0.001, coef0=1.0, cost=0.1),  # the reference's kernel-test parameters (generic_csvm_tests.hpp:372-493)
    "def": dict(degree=3, gamma=None, coef0=0.0, cost=1.0),   # csvm defaults, gamma = 1 / num_features
}
KERNELS = ["linear", "polynomial", "rbf"]
DATASETS = ["5x4", "blobs263x37", "500x200"]
DTYPES = {"f32": np.float32, "f64": np.float64}


def pytest_collection_modifyitems(config, items):
    """Tests that import torch (and create an RCCL communicator) run last: see the OpenMP note above."""
    late = [it for it in items if "test_gpu_interop" in it.nodeid]
    if late:
        items[:] = [it for it in items if "test_gpu_interop" not in it.nodeid] + late


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _library_options_are_restored():
    """A test may change the library's process-wide option defaults; whatever it leaves behind is put back, so that no later test
    silently runs a non-default kernel (round 1: a test left gram_mode at 0 for everything after it)."""
    from plssvm_amd import _capi

    before = {name: _capi.get_option(name) for name in _capi.OPTION_NAMES}
    yield
    for name, value in before.items():
        if _capi.get_option(name) != value:
            _capi.set_option(name, value)


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(HERE, "golden", "golden.npz"))


@pytest.fixture(scope="session")
def inputs():
    return np.load(os.path.join(HERE, "golden", "inputs.npz"))


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.oracle()


def resolved_kw(P, d):
    return dict(degree=P["degree"], gamma=(P["gamma"] if P["gamma"] is not None else 1.0 / d), coef0=P["coef0"])
