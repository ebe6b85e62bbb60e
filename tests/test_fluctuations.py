"""LocalFluctuations (lib/visfd/filter3d.hpp:1698-1926): CPU restatement against the real reference
and the committed golden vectors; the HIP path (two Gaussians + two element-wise kernels) against the restatement
through the C ABI, bit for bit."""
import os

import numpy as np
import pytest

import volgen
from conftest import assert_bits_equal

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fluctuations.npz")

# (shape [nz,ny,nx], radius (x,y,z), truncate ratio (<0: from the 0.03 threshold), masked, normalize)
CASES = [
    ((20, 24, 28), (4.0, 4.0, 4.0), -1.0, False, True),
    ((20, 24, 28), (4.0, 4.0, 4.0), -1.0, True, True),
    ((17, 21, 36), (3.0, 5.0, 2.5), 2.5, False, True),      # anisotropic, explicit window
    ((17, 21, 36), (6.0, 6.0, 6.0), 2.0, False, False),     # no boundary normalisation
    ((9, 40, 44), (8.0, 8.0, 8.0), -1.0, True, False),      # window wider than the image in z
]


def case_inputs(i):
    shape, radius, ratio, masked, norm = CASES[i]
    src = volgen.noise_volume(shape, seed=500 + i)
    mask = volgen.block_mask(shape, seed=600 + i) if masked else None
    return src, mask, radius, ratio, norm


def sigmas(radius, ratio):
    from visfd_amd import api
    return api.fluctuation_sigmas(radius, 2.0, ratio, 0.03)


@pytest.fixture(scope="module")
def ctx():
    from visfd_amd import api
    c = api.Context(0)
    yield c
    c.close()


def test_fluctuation_parameters():
    """sigma = radius / (9 pi / 2)^(1/6); the default window follows from the 0.03 decay threshold as
    sqrt(-log 0.03) -- not the Gaussian filters' sqrt(-2 log 0.03) (filter3d_variants.hpp:663-669)."""
    sg, ratio = sigmas((4.0, 8.0, 2.0), -1.0)
    k = np.float32(np.power(4.5 * np.pi, 1.0 / 6.0))
    assert sg == tuple(float(np.float32(r) / k) for r in (4.0, 8.0, 2.0))
    assert ratio == float(np.float32(np.sqrt(np.float64(-np.log(np.float32(0.03))))))
    assert sigmas((4.0, 4.0, 4.0), 2.5)[1] == 2.5


def test_oracle_fluctuations_golden(oracle):
    g = np.load(GOLD)
    for i in range(len(CASES)):
        src, mask, radius, ratio, norm = case_inputs(i)
        sg, r = sigmas(radius, ratio)
        assert_bits_equal(oracle.local_fluctuations(src, sg, r, mask, norm), g["case%d" % i], "fluctuations case %d" % i)


def test_oracle_fluctuations_vs_reference(oracle, ref):
    for i in range(len(CASES)):
        src, mask, radius, ratio, norm = case_inputs(i)
        sg, r = sigmas(radius, ratio)
        assert_bits_equal(oracle.local_fluctuations(src, sg, r, mask, norm),
                          ref.local_fluctuations(src, sg, r, mask, norm), "fluctuations case %d" % i)
    rng = np.random.default_rng(3)
    for k in range(4):
        shape = tuple(int(rng.integers(8, 26)) for _ in range(3))
        src = volgen.noise_volume(shape, 700 + k)
        sg = tuple(float(rng.uniform(0.7, 3.0)) for _ in range(3))
        r = float(rng.uniform(1.5, 2.8))
        assert_bits_equal(oracle.local_fluctuations(src, sg, r, None, bool(k & 1)),
                          ref.local_fluctuations(src, sg, r, None, bool(k & 1)), "random fluctuations %d" % k)


@pytest.mark.gpu
def test_gpu_fluctuations_parity(ctx, oracle):
    from visfd_amd import api
    g = np.load(GOLD)
    for i in range(len(CASES)):
        src, mask, radius, ratio, norm = case_inputs(i)
        sg, r = sigmas(radius, ratio)
        got = ctx.local_fluctuations(src, sg, r, mask, norm)
        assert_bits_equal(got, oracle.local_fluctuations(src, sg, r, mask, norm), "fluctuations case %d" % i)
        assert_bits_equal(got, g["case%d" % i], "fluctuations case %d vs golden" % i)
    # a shape the single-sweep Gaussian takes (nx a multiple of 4), constant image -> zero fluctuation
    src = volgen.noise_volume((24, 40, 72), seed=9)
    sg, r = sigmas((5.0, 5.0, 5.0), -1.0)
    assert_bits_equal(ctx.local_fluctuations(src, sg, r), oracle.local_fluctuations(src, sg, r), "single-sweep shape")
    flat = np.full((12, 16, 20), 3.5, np.float32)
    out = ctx.local_fluctuations(flat, sg, r)
    assert_bits_equal(out, oracle.local_fluctuations(flat, sg, r), "constant image")
    with pytest.raises(api.VisfdHipError):
        ctx.local_fluctuations(src, sg, r, None, True, 4.0)      # generalised Gaussians are not provided
