"""The CPU restatement (oracle/) against the golden vectors produced by the real reference.

This is what pins the oracle: every comparison is BIT-EXACT (the restatement keeps the
reference's operand order; both are plain IEEE float/double without FMA).
"""
import os

import numpy as np
import pytest

import volgen
from conftest import GOLDEN, assert_bits_equal, golden
from oracle import pyoracle as po


def test_taps(oracle):
    g = golden("taps")
    assert np.float32(oracle.ratio_from_threshold(0.03)) == g["ratio"]
    for s, h in volgen.TAP_CASES:
        assert_bits_equal(oracle.gauss_taps(s, h), g["s%g_h%d" % (s, h)], "taps s=%g h=%d" % (s, h))


def test_gauss_on_reference_fixture(oracle):
    """BASELINE config 1: filter_mrc -gauss 2 -w 1 on tests/test_blob_detect.rec."""
    g = golden("gauss_blobrec")
    img = volgen.read_mrc(os.path.join(GOLDEN, "test_blob_detect.rec"))
    msk = volgen.read_mrc(os.path.join(GOLDEN, "test_blob_detect_mask.rec"))
    ratio = oracle.ratio_from_threshold(0.03)
    out, A = oracle.gauss_ratio(img, (2, 2, 2), ratio)
    assert_bits_equal(out, g["out"], "gauss")
    assert np.float32(A) == g["A"]
    # known answers recorded in SURVEY.md §4 (printed by the reference CLI)
    assert abs(A - 0.00907605) < 1e-8
    assert abs(float(out.min()) - 33.43096) < 1e-4 and abs(float(out.max()) - 41.48531) < 1e-4
    assert abs(float(out.mean(dtype=np.float64)) - 36.50518) < 1e-4
    out, A = oracle.gauss_ratio(img, (2, 2, 2), ratio, msk)
    assert_bits_equal(out, g["out_masked"], "gauss masked")


def test_gauss_dog_log_seeded(oracle):
    g = golden("gauss_seeded")
    ratio = oracle.ratio_from_threshold(0.03)
    src = volgen.noise_volume(volgen.GAUSS_SHAPE, seed=101)
    mask = volgen.block_mask(volgen.GAUSS_SHAPE, seed=102)
    for tag, m in (("nomask", None), ("mask", mask)):
        for norm in (True, False):
            o, A = oracle.gauss_hw(src, volgen.ANISO_SIGMA, volgen.ANISO_HW, m, norm)
            assert_bits_equal(o, g["aniso_%s_norm%d" % (tag, norm)], "aniso %s %d" % (tag, norm))
            assert np.float32(A) == g["aniso_%s_norm%d_A" % (tag, norm)]
    o, _ = oracle.gauss_hw(np.ascontiguousarray(src[:4, :5, :3]), (2, 2, 2), (5, 5, 5))
    assert_bits_equal(o, g["tiny_n_lt_window"], "tiny")
    o, A, B = oracle.log(src, (2, 2, 2), 0.02, ratio)
    assert_bits_equal(o, g["log_nomask"], "log")
    assert_bits_equal(np.array([A, B], np.float32), g["log_AB"], "log A,B")
    o, _, _ = oracle.log(src, (2.5, 2, 1.5), 0.02, ratio, mask)
    assert_bits_equal(o, g["log_mask_aniso"], "log masked aniso")
    o, _, _ = oracle.dog(src, (1.5,) * 3, (2.5,) * 3, (6, 6, 6))
    assert_bits_equal(o, g["dog_nomask"], "dog")


def test_blob_reference_test_command(oracle):
    """tests/test_blob_detection.sh:21 of the reference: 58 scales, masked; 11 minima, best
    blob written as '235.2 392 313.6 177.915 -140.018' (SURVEY.md §4)."""
    g = golden("blob_rec")
    img = volgen.read_mrc(os.path.join(GOLDEN, "test_blob_detect.rec"))
    msk = volgen.read_mrc(os.path.join(GOLDEN, "test_blob_detect_mask.rec"))
    diam = volgen.cli_blob_diameters(160.0, 280.0, 1.01, 1.0) / np.float32(19.6)
    assert len(diam) == 58
    assert_bits_equal(diam, g["diam_vox"], "diameter ladder")
    sig = oracle.diameters_to_sigmas(diam)
    assert_bits_equal(sig, g["sigmas"], "sigmas")
    ratio = oracle.ratio_from_threshold(0.03)
    mins, maxs = oracle.blob_dog(img, sig, msk, None, 0.02, ratio, 0.0, -np.inf, False)
    mins = volgen.sort_blobs(mins, True)
    assert len(mins) == 11
    assert_bits_equal(mins, g["minima"], "minima")
    d = oracle.sigmas_to_diameters(np.ascontiguousarray(mins[:, 3]))
    assert_bits_equal(d, g["minima_diam_vox"], "diameters")
    w = np.float32(19.6)
    line = "%g %g %g %g %g" % (mins[0, 0] * w, mins[0, 1] * w, mins[0, 2] * w, d[0] * w, mins[0, 4])
    assert line == "235.2 392 313.6 177.915 -140.018"


def test_blob_seeded(oracle):
    g = golden("blob_seeded")
    ratio = oracle.ratio_from_threshold(0.03)
    src = volgen.blob_volume(volgen.BLOB_SHAPE, seed=201)
    mask = volgen.block_mask(volgen.BLOB_SHAPE, seed=202)
    sig = oracle.diameters_to_sigmas(volgen.BLOB_DIAMS)
    assert_bits_equal(sig, g["sigmas"], "sigmas")
    total = 0
    for tag, m in (("nomask", None), ("mask", mask)):
        for mode, kw in volgen.BLOB_MODES.items():
            a, b = oracle.blob_dog(src, sig, m, None, 0.02, ratio, **kw)
            assert_bits_equal(volgen.sort_blobs(a, True), g["%s_%s_min" % (tag, mode)], tag + mode + " min")
            assert_bits_equal(volgen.sort_blobs(b, False), g["%s_%s_max" % (tag, mode)], tag + mode + " max")
            total += len(a) + len(b)
    assert total > 10


@pytest.mark.parametrize("oname,order", [("dec", po.ORDER_DECREASING), ("inc", po.ORDER_INCREASING)])
def test_membrane_reference_fixture(oracle, oname, order):
    g = golden("membrane_rec")
    mem = volgen.read_mrc(os.path.join(GOLDEN, "test_image_membrane.rec"))
    ratio = oracle.ratio_from_threshold(0.03)
    sigma = np.float32(1.5)
    grad, hess = oracle.calc_hessian(mem, sigma, ratio)
    assert_bits_equal(hess, g["hess"], "hessian")
    assert_bits_equal(grad, g["grad"], "gradient")
    sal, dirs = oracle.hessian_saliency(hess, order)
    assert_bits_equal(sal, g["sal_" + oname], "saliency")
    assert_bits_equal(dirs, g["dir_" + oname], "direction")
    thr = oracle.threshold_fraction(sal, 0.1)
    assert np.float32(thr) == g["thr_" + oname]
    assert_bits_equal(sal, g["salthr_" + oname], "thresholded saliency")
    ten = oracle.tv_dense_stick(sal, dirs, 4 * sigma / 2, 4, 2.0 ** 0.5)
    assert_bits_equal(ten, g["tensor_" + oname], "vote tensor")
    s2 = sal.copy()
    oracle.tensor_saliency(ten, order, s2)
    assert_bits_equal(s2, g["tvsal_" + oname], "post-TV saliency")


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_membrane_seeded(oracle, tag):
    g = golden("membrane_seeded")
    ratio = oracle.ratio_from_threshold(0.03)
    src = volgen.membrane_volume(volgen.MEM_SHAPE, seed=301)
    m = volgen.block_mask(volgen.MEM_SHAPE, seed=302) if tag == "mask" else None
    grad, hess = oracle.calc_hessian(src, volgen.MEM_SIGMA, ratio, m)
    assert_bits_equal(hess, g[tag + "_hess"], "hessian")
    assert_bits_equal(grad, g[tag + "_grad"], "gradient")
    sal, dirs = oracle.hessian_saliency(hess, po.ORDER_DECREASING, m)
    assert_bits_equal(sal, g[tag + "_sal"], "saliency")
    assert_bits_equal(dirs, g[tag + "_dir"], "direction")
    thr = oracle.threshold_fraction(sal, volgen.MEM_FRACTION, m)
    assert np.float32(thr) == g[tag + "_thr"]
    assert_bits_equal(sal, g[tag + "_salthr"], "thresholded")
    for ex in (4, 2, 3):
        ten = oracle.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, ex, 2.0 ** 0.5, m, m)
        assert_bits_equal(ten, g["%s_tensor_e%d" % (tag, ex)], "tensor e%d" % ex)
    ten = oracle.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, 4, 2.0 ** 0.5, m, m)
    s2 = sal.copy()
    oracle.tensor_saliency(ten, po.ORDER_DECREASING, s2, m)
    assert_bits_equal(s2, g[tag + "_tvsal"], "post-TV saliency")
    ten = oracle.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, 4, 2.0 ** 0.5, m, m, curves=True)
    assert_bits_equal(ten, g[tag + "_tensor_curves"], "curve-mode tensor")


def test_tv_tables(oracle):
    g = golden("membrane_seeded")
    h, w, rh = oracle.tv_tables(8.66, 2.0 ** 0.5)
    assert h == 12
    assert_bits_equal(w, g["tvtab_h12_w"], "w h=12")
    h, w, rh = oracle.tv_tables(volgen.MEM_TV_SIGMA, 2.0 ** 0.5)
    assert_bits_equal(w, g["tvtab_w"], "w")
    assert_bits_equal(rh, g["tvtab_rhat"], "rhat")


def test_eigen(oracle):
    g = golden("eigen")
    mats = volgen.eigen_cases(seed=401)
    assert_bits_equal(mats, g["mats"], "inputs")
    for oname, order in (("inc", 0), ("dec", 1)):
        assert_bits_equal(oracle.diagonalize(mats, order), g["diag_" + oname], "diag " + oname)
        ev, evec = oracle.evects(mats, order)
        assert_bits_equal(ev, g["evals_" + oname], "evals")
        assert_bits_equal(evec, g["evecs_" + oname], "evecs")
