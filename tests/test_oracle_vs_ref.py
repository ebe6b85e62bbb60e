"""Direct oracle-vs-reference checks on fresh random inputs (only where oracle/_ref was built,
i.e. in the container that has /root/reference; skipped elsewhere)."""
import numpy as np
import pytest

import volgen
from conftest import assert_bits_equal


@pytest.mark.parametrize("seed", [11, 12])
def test_filters(oracle, ref, seed):
    rng = np.random.default_rng(seed)
    shape = tuple(int(rng.integers(9, 30)) for _ in range(3))
    src = volgen.noise_volume(shape, seed)
    mask = volgen.block_mask(shape, seed + 1)
    sig = tuple(float(rng.uniform(0.6, 3.0)) for _ in range(3))
    ratio = oracle.ratio_from_threshold(0.03)
    for m in (None, mask):
        for norm in (True, False):
            a, A = oracle.gauss_ratio(src, sig, ratio, m, norm)
            b, B = ref.gauss_ratio(src, sig, ratio, m, norm)
            assert_bits_equal(a, b, "gauss")
            assert A == B
        a = oracle.log(src, sig, 0.02, ratio, m)
        b = ref.log(src, sig, 0.02, ratio, m)
        assert_bits_equal(a[0], b[0], "log")
        assert a[1:] == b[1:]


def test_sparse_inputs_hit_the_zero_skip(oracle, ref):
    """Mostly-zero images exercise the reference's sparse-input shortcut (filter1d.hpp:59-94)."""
    rng = np.random.default_rng(5)
    src = np.zeros((18, 20, 22), np.float32)
    idx = rng.integers(0, src.size, 12)
    src.reshape(-1)[idx] = rng.standard_normal(12).astype(np.float32) * 50
    src[3, 4, 5] = -0.0
    a, _ = oracle.gauss_hw(src, (1.2, 1.2, 1.2), (3, 3, 3), None, False)
    b, _ = ref.gauss_hw(src, (1.2, 1.2, 1.2), (3, 3, 3), None, False)
    assert_bits_equal(a, b, "sparse gauss")
    mask = np.zeros_like(src)
    mask[5:12, 6:14, 7:15] = 1
    a, _ = oracle.gauss_hw(src + 7, (1.2, 1.2, 1.2), (3, 3, 3), mask, True)
    b, _ = ref.gauss_hw(src + 7, (1.2, 1.2, 1.2), (3, 3, 3), mask, True)
    assert_bits_equal(a, b, "sparse-mask gauss")


def test_blobs_and_ridges(oracle, ref):
    src = volgen.blob_volume((22, 25, 27), seed=77)
    mask = volgen.block_mask(src.shape, seed=78)
    sig = oracle.diameters_to_sigmas(np.array([4.5, 5.5, 6.5, 7.5, 9], np.float32))
    for m in (None, mask):
        for kw in volgen.BLOB_MODES.values():
            a = oracle.blob_dog(src, sig, m, None, 0.02, 2.5, **kw)
            b = ref.blob_dog(src, sig, m, None, 0.02, 2.5, **kw)
            for x, y, asc in ((a[0], b[0], True), (a[1], b[1], False)):
                assert_bits_equal(volgen.sort_blobs(x, asc), volgen.sort_blobs(y, asc), "blobs")
        ga, ha = oracle.calc_hessian(src, 1.4, 2.5, m)
        gb, hb = ref.calc_hessian(src, 1.4, 2.5, m)
        assert_bits_equal(ga, gb, "grad")
        assert_bits_equal(ha, hb, "hess")
        for order in (0, 1):
            sa = oracle.hessian_saliency(ha, order, m)
            sb = ref.hessian_saliency(hb, order, m)
            assert_bits_equal(sa[0], sb[0], "sal")
            assert_bits_equal(sa[1], sb[1], "dir")
            s1, s2 = sa[0].copy(), sb[0].copy()
            assert oracle.threshold_fraction(s1, 0.07, m) == ref.threshold_fraction(s2, 0.07, m)
            assert_bits_equal(s1, s2, "thr")
            ta = oracle.tv_dense_stick(s1, sa[1], 2.6, 4, 2 ** 0.5, m, m)
            tb = ref.tv_dense_stick(s2, sb[1], 2.6, 4, 2 ** 0.5, m, m)
            assert_bits_equal(ta, tb, "tv")
            oracle.tensor_saliency(ta, order, s1, m)
            ref.tensor_saliency(tb, order, s2, m)
            assert_bits_equal(s1, s2, "tv sal")


def test_eigen_random(oracle, ref):
    mats = volgen.eigen_cases(seed=9, nrand=20000)
    for order in (0, 1):
        assert_bits_equal(oracle.diagonalize(mats, order), ref.diagonalize(mats, order), "diag")
        a, b = oracle.evects(mats, order), ref.evects(mats, order)
        assert_bits_equal(a[0], b[0], "evals")
        assert_bits_equal(a[1], b[1], "evecs")
