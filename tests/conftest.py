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


PERVOXEL_LOG = os.path.join(ROOT, "gpurun_out", "tolerance_pervoxel.jsonl")


def pervoxel_stats(a, b, rtol=1e-5, floor=1e-3):
    """The per-voxel reading of "within 1e-5 relative": among the SIGNIFICANT values of b (|b| > floor * max|b|), the fraction
    with |a-b| > rtol * |b|, and the largest |a-b| / |b| among them.  (The field-scale reading is assert_close_rel's.)"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = float(np.max(np.abs(b))) if b.size else 0.0
    sig = np.abs(b) > floor * scale
    n = int(sig.sum())
    if n == 0:
        return {"significant": 0, "frac_over": 0.0, "max_rel": 0.0, "scale": scale}
    rel = np.abs(a[sig] - b[sig]) / np.abs(b[sig])
    return {"significant": n, "frac_over": float(np.mean(rel > rtol)), "max_rel": float(rel.max()), "scale": scale,
            "max_err_of_scale": float(np.max(np.abs(a - b)) / scale)}


def assert_close_rel(a, b, rtol=1e-5, what="", pervoxel=None):
    """|a-b| <= rtol * max|b| (relative to the field's scale): the float tolerance of BASELINE.json's north_star ("within
    1e-5 relative for float voxel values") in its field-scale reading.  pervoxel = f adds the stricter per-voxel reading as a
    bound: at most the fraction f of the significant voxels (|b| > 1e-3 max|b|) may have |a-b| > rtol * |b|; the measured
    fraction and the largest per-voxel relative error are appended to gpurun_out/tolerance_pervoxel.jsonl either way."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(np.max(np.abs(b))) if b.size else 0.0
    err = float(np.max(np.abs(a - b))) if b.size else 0.0
    assert err <= rtol * scale + 1e-30, "%s: max|a-b|=%g > %g * scale(%g)" % (what, err, rtol, scale)
    if pervoxel is not None:
        st = pervoxel_stats(a, b, rtol)
        try:
            import json
            os.makedirs(os.path.dirname(PERVOXEL_LOG), exist_ok=True)
            with open(PERVOXEL_LOG, "a") as f:
                f.write(json.dumps(dict(what=what, bound=pervoxel, **st)) + "\n")
        except OSError:
            pass
        assert st["frac_over"] <= pervoxel, "%s: %.3g of the %d significant voxels differ by more than %g of their own value (bound %g; worst %.3g)" % (
            what, st["frac_over"], st["significant"], rtol, pervoxel, st["max_rel"])
