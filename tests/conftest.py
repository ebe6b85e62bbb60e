import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """The tests call the product through visfd_amd/libvisfd_hip.so; build it (hipcc cross-compiles without a GPU)
    when a fresh checkout has not run `python -m visfd_amd.build` / __graft_entry__.build() yet."""
    from visfd_amd import api
    if not os.path.exists(api.LIB_PATH):
        from visfd_amd import build
        build.build(verbose=False)


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/libvisfd_oracle.so); built on demand with g++."""
    from oracle import pyoracle as po
    if not po.available("oracle"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libvisfd_oracle.so"])
    return po.load("oracle")


@pytest.fixture(scope="session")
def ref():
    """The real reference templates (oracle/_ref/libvisfd_ref.so) when that build exists."""
    from oracle import pyoracle as po
    if not po.available("ref"):
        pytest.skip("oracle/_ref/libvisfd_ref.so not built (needs /root/reference)")
    return po.load("ref")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert a.dtype == b.dtype, (what, a.dtype, b.dtype)
    if a.dtype == np.float32:
        ne = a.view(np.uint32) != b.view(np.uint32)
    else:
        ne = a != b
    n = int(ne.sum())
    if n:
        idx = np.argwhere(ne)[0]
        raise AssertionError("%s: %d of %d values differ bitwise; first at %s: %r vs %r" % (
            what, n, a.size, tuple(idx), a[tuple(idx)], b[tuple(idx)]))


def assert_close_rel(a, b, rtol=1e-5, what=""):
    """|a-b| <= rtol * max|b| (relative to the field's scale): the float tolerance of
    BASELINE.json's north_star ("within 1e-5 relative for float voxel values")."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(np.max(np.abs(b))) if b.size else 0.0
    err = float(np.max(np.abs(a - b))) if b.size else 0.0
    assert err <= rtol * scale + 1e-30, "%s: max|a-b|=%g > %g * scale(%g)" % (what, err, rtol, scale)
