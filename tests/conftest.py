import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Parity bar of BASELINE.json: max|a-b| / max|ref| <= 1e-6 in float64.  The HIP path actually lands ~1e-13.
REL_TOL = 1e-6


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(a, ref):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    scale = max(float(np.max(np.abs(ref))), 1e-300)
    return float(np.max(np.abs(a - ref))) / scale


def assert_close(a, ref, tol=REL_TOL, what=""):
    e = rel_err(a, ref)
    assert e <= tol, f"{what}: relative error {e:.3e} > {tol:.1e}"


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu
    return cpu.oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference (only where oracle/_ref was built, i.e. where /root/reference exists or the .so travelled)."""
    from oracle import cpu
    if not cpu.have_reference():
        pytest.skip("oracle/_ref/libproxtv_ref.so not built here")
    return cpu.reference()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def g1d():
    return load_golden("golden_1d.npz")


@pytest.fixture(scope="session")
def g2d():
    return load_golden("golden_2d.npz")


@pytest.fixture(scope="session")
def g1dm():
    return load_golden("golden_1d_other_methods.npz")


@pytest.fixture(scope="session")
def gpd():
    return load_golden("golden_2d_primal_dual.npz")


@pytest.fixture(scope="session")
def gnd():
    return load_golden("golden_nd.npz")


@pytest.fixture(scope="session")
def glarge():
    return load_golden("golden_large.npz")


@pytest.fixture(scope="session")
def ptv():
    """The product's Python surface, with a usable device (gpu tests only)."""
    import proxtv_amd
    from proxtv_amd import _lib
    _lib.require_device()
    return proxtv_amd


@pytest.fixture(scope="session")
def clib():
    """The product's C-ABI through ctypes (gpu tests call through this)."""
    from proxtv_amd import _lib
    return _lib.require_device()
