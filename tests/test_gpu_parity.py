"""Parity of the HIP path (through the C ABI, host-pointer face) against the CPU oracle, the
committed golden vectors of the real reference, and the reference's own test answers.

Bars (BASELINE.json north_star): bit-exact for indices (blob lists, thresholds' selected sets) and
for every stage whose arithmetic is plain IEEE float (Gaussian, DoG/LoG, Hessian, tensor voting
with exponent 2/4); 1e-5 relative (to the field's scale) where device libm enters (eigen solver:
atan2/sin/cos, pow for odd exponents)."""
import os

import numpy as np
import pytest

import volgen
from conftest import GOLDEN, assert_bits_equal, assert_close_rel, golden
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from visfd_amd import api
    c = api.Context(0)
    yield c
    c.close()


RATIO = None


def ratio(oracle):
    return oracle.ratio_from_threshold(0.03)


# ------------------------------------------------------------------------------------------ Gaussian
def test_gauss_reference_fixture(ctx, oracle):
    g = golden("gauss_blobrec")
    img = volgen.read_mrc(os.path.join(GOLDEN, "test_blob_detect.rec"))
    msk = volgen.read_mrc(os.path.join(GOLDEN, "test_blob_detect_mask.rec"))
    out, A = ctx.gauss_ratio(img, (2, 2, 2), ratio(oracle))
    assert_bits_equal(out, g["out"], "gauss on test_blob_detect.rec")
    assert np.float32(A) == g["A"]
    out, A = ctx.gauss_ratio(img, (2, 2, 2), ratio(oracle), msk)
    assert_bits_equal(out, g["out_masked"], "masked gauss on test_blob_detect.rec")


def test_gauss_dog_log_seeded(ctx, oracle):
    g = golden("gauss_seeded")
    r = ratio(oracle)
    src = volgen.noise_volume(volgen.GAUSS_SHAPE, seed=101)
    mask = volgen.block_mask(volgen.GAUSS_SHAPE, seed=102)
    for tag, m in (("nomask", None), ("mask", mask)):
        for norm in (True, False):
            o, A = ctx.gauss_hw(src, volgen.ANISO_SIGMA, volgen.ANISO_HW, m, norm)
            assert_bits_equal(o, g["aniso_%s_norm%d" % (tag, norm)], "aniso %s %d" % (tag, norm))
            assert np.float32(A) == g["aniso_%s_norm%d_A" % (tag, norm)]
    o, _ = ctx.gauss_hw(np.ascontiguousarray(src[:4, :5, :3]), (2, 2, 2), (5, 5, 5))
    assert_bits_equal(o, g["tiny_n_lt_window"], "image smaller than the window")
    o, A, B = ctx.log(src, (2, 2, 2), 0.02, r)
    assert_bits_equal(o, g["log_nomask"], "log")
    assert_bits_equal(np.array([A, B], np.float32), g["log_AB"], "log A,B")
    o, _, _ = ctx.log(src, (2.5, 2, 1.5), 0.02, r, mask)
    assert_bits_equal(o, g["log_mask_aniso"], "log masked aniso")
    o, _, _ = ctx.dog(src, (1.5,) * 3, (2.5,) * 3, (6, 6, 6))
    assert_bits_equal(o, g["dog_nomask"], "dog")


@pytest.mark.parametrize("shape", [(40, 50, 70), (33, 17, 129), (7, 9, 200), (64, 64, 64), (20, 45, 136)])
@pytest.mark.parametrize("h", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12])
def test_gauss_fused_vs_oracle(ctx, oracle, shape, h):
    """Every single-sweep instantiation (isotropic window h=1..8; nx a multiple of 4) and the 3-pass path (other
    widths, h > 8; odd nx for h = 9, 10, the widest windows of the bench's blob scales), on shapes with ragged tiles,
    against the oracle, with and without normalisation."""
    if h > 6 and shape not in ((40, 50, 70), (20, 45, 136)) and not (h in (9, 10) and shape == (33, 17, 129)):
        pytest.skip("wide windows: two shapes (3-pass and single-sweep) are enough")
    src = volgen.noise_volume(shape, seed=1000 + h)
    sigma = (h / 2.6,) * 3
    for norm in (True, False):
        a, A = ctx.gauss_hw(src, sigma, (h, h, h), None, norm)
        b, B = oracle.gauss_hw(src, sigma, (h, h, h), None, norm)
        assert_bits_equal(a, b, "gauss h=%d shape=%s norm=%d" % (h, shape, norm))
        assert A == B


@pytest.mark.parametrize("h", [9, 10])
def test_dog_wide_windows_vs_oracle(ctx, oracle, h):
    """DoG whose Gaussians are too wide for the single sweep (h = 9, 10: three single-axis kernels, the X pass subtracts
    from the first Gaussian's result as it stores), masked and unmasked, odd nx."""
    shape = (36, 41, 75)
    src = volgen.noise_volume(shape, seed=400 + h)
    mask = (np.random.default_rng(h).random(shape) > 0.2).astype(np.float32)
    sa, sb = (h / 3.1,) * 3, (h / 2.6,) * 3
    for m in (None, mask):
        a = ctx.dog(src, sa, sb, (h, h, h), m)
        b = oracle.dog(src, sa, sb, (h, h, h), m)
        assert_bits_equal(a[0] if isinstance(a, tuple) else a, b[0] if isinstance(b, tuple) else b, "dog h=%d mask=%s" % (h, m is not None))


@pytest.mark.parametrize("h", [2, 5, 7, 9])
def test_gauss_fused_extreme_magnitudes(ctx, oracle, h):
    """The single-sweep kernel divides by the boundary normaliser through a reciprocal with exact residual
    corrections where that is provably the IEEE quotient and through the full-range division elsewhere (zeros,
    denormals, huge and tiny magnitudes, the first/last h planes): every mixture must give the reference's bits."""
    rng = np.random.default_rng(900 + h)
    shape = (48, 40, 136)
    n = int(np.prod(shape))
    cases = {
        "wide": (rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-37.0, 37.0, n)),
        "tiny": (rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-44.0, -28.0, n)),       # denormals included
        "huge": (rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(28.0, 37.5, n)),
        "sparse": np.where(rng.random(n) < 0.97, 0.0, rng.normal(0.0, 50.0, n)),
        "neg": -np.abs(rng.normal(1000.0, 100.0, n)),
    }
    sigma = (h / 2.6,) * 3
    with np.errstate(over="ignore", under="ignore"):
        for name, v in cases.items():
            src = np.ascontiguousarray(v.astype(np.float32).reshape(shape))
            src[~np.isfinite(src)] = 0.0
            a, _ = ctx.gauss_hw(src, sigma, (h, h, h), None, True)
            b, _ = oracle.gauss_hw(src, sigma, (h, h, h), None, True)
            fin = np.isfinite(b)    # sums of huge values may overflow: NaN/Inf voxels are out of contract
            assert_bits_equal(np.where(fin, a, 0).astype(np.float32), np.where(fin, b, 0).astype(np.float32),
                              "gauss h=%d %s magnitudes" % (h, name))


def _oracle_separable(oracle, src, taps, normalize):
    import ctypes as C
    fp = C.POINTER(C.c_float)
    nz, ny, nx = src.shape
    out = np.empty_like(src)
    oracle.lib.vo_separable3d.restype = C.c_float
    oracle.lib.vo_separable3d.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, fp,
                                          C.c_int, C.c_int]
    h = [(len(t) - 1) // 2 for t in taps]
    A = oracle.lib.vo_separable3d(src.ctypes.data_as(fp), out.ctypes.data_as(fp), None, nx, ny, nz,
                                  taps[0].ctypes.data_as(fp), h[0], taps[1].ctypes.data_as(fp), h[1],
                                  taps[2].ctypes.data_as(fp), h[2], int(normalize))
    return out, A


def test_separable_symmetric_signed_taps(ctx, oracle):
    """Symmetric taps with negative lobes and zero samples through the single-sweep kernel (its Z pass shares the
    products of the taps +j and -j); Z taps that are not symmetric take the three single-axis kernels; a positive
    non-Gaussian window also with the boundary normaliser."""
    rng = np.random.default_rng(31)
    shape = (30, 44, 72)
    src = volgen.noise_volume(shape, seed=77)
    src[rng.random(shape) < 0.3] = 0.0
    for h in (3, 6):
        half = rng.normal(0.0, 1.0, h + 1).astype(np.float32)
        sym = np.ascontiguousarray(np.concatenate([half[:0:-1], half]), np.float32)
        asym = rng.normal(0.0, 1.0, 2 * h + 1).astype(np.float32)
        tri = np.ascontiguousarray(np.concatenate([np.arange(1, h + 1), [h + 1], np.arange(h, 0, -1)]), np.float32)
        tri /= np.float32(tri.sum() * 1.01)
        for taps, norm in (([sym, asym, sym], False), ([sym, sym, asym], False), ([tri, sym, tri], False),
                           ([tri, tri, tri], True)):
            a, A = ctx.separable3d(src, taps, None, norm)
            b, B = _oracle_separable(oracle, src, taps, norm)
            assert_bits_equal(a, b, "generic taps h=%d norm=%d" % (h, norm))
            assert A == B


def test_gauss_masked_vs_oracle(ctx, oracle):
    shape = (30, 41, 67)
    src = volgen.noise_volume(shape, seed=5)
    mask = volgen.block_mask(shape, seed=6)
    wmask = mask * np.float32(0.5) + np.float32(0.25) * (volgen.noise_volume(shape, 7, 0.5, 0.1) > 0.5)
    for m in (mask, np.ascontiguousarray(wmask, np.float32)):
        for norm in (True, False):
            a, _ = ctx.gauss_hw(src, (1.7, 2.2, 1.1), (4, 5, 2), m, norm)
            b, _ = oracle.gauss_hw(src, (1.7, 2.2, 1.1), (4, 5, 2), m, norm)
            assert_bits_equal(a, b, "masked gauss norm=%d" % norm)


def test_gauss_sparse_and_empty_mask(ctx, oracle):
    rng = np.random.default_rng(5)
    src = np.zeros((18, 20, 22), np.float32)
    idx = rng.integers(0, src.size, 12)
    src.reshape(-1)[idx] = rng.standard_normal(12).astype(np.float32) * 50
    src[3, 4, 5] = -0.0
    a, _ = ctx.gauss_hw(src, (1.2,) * 3, (3, 3, 3), None, False)
    b, _ = oracle.gauss_hw(src, (1.2,) * 3, (3, 3, 3), None, False)
    assert_bits_equal(a, b, "sparse gauss")
    mask = np.zeros_like(src)
    mask[5:12, 6:14, 7:15] = 1
    a, _ = ctx.gauss_hw(src + 7, (1.2,) * 3, (3, 3, 3), mask, True)
    b, _ = oracle.gauss_hw(src + 7, (1.2,) * 3, (3, 3, 3), mask, True)
    assert_bits_equal(a, b, "mostly-zero mask")
    a, _ = ctx.gauss_hw(src + 7, (1.2,) * 3, (3, 3, 3), np.zeros_like(src), True)
    b, _ = oracle.gauss_hw(src + 7, (1.2,) * 3, (3, 3, 3), np.zeros_like(src), True)
    assert_bits_equal(a, b, "all-zero mask")


@pytest.mark.parametrize("h", [2, 5])
def test_gauss_fused_signed_zeros(ctx, oracle, h):
    """The single-sweep kernel starts its sums from the first product instead of from +0.0 and adds +0.0 once at the
    end: regions of -0.0 samples (alone, next to +0.0, next to data, under negative taps) must still come out with the
    reference's zero signs."""
    rng = np.random.default_rng(40 + h)
    shape = (24, 36, 72)
    src = np.full(shape, -0.0, np.float32)
    src[:, :, 36:] = 0.0
    src[8:12, 10:20, 20:50] = rng.standard_normal((4, 10, 30)).astype(np.float32)
    src[rng.random(shape) < 0.02] = np.float32(-3.5)
    sigma = (h / 2.6,) * 3
    for norm in (True, False):
        a, _ = ctx.gauss_hw(src, sigma, (h, h, h), None, norm)
        b, _ = oracle.gauss_hw(src, sigma, (h, h, h), None, norm)
        assert np.signbit(src).any() and (b == 0).any()
        assert_bits_equal(a, b, "signed zeros h=%d norm=%d" % (h, norm))
    half = rng.normal(0.0, 1.0, h + 1).astype(np.float32)
    sym = np.ascontiguousarray(np.concatenate([half[:0:-1], half]), np.float32)
    a, _ = ctx.separable3d(src, [sym, -np.abs(sym), sym], None, False)
    b, _ = _oracle_separable(oracle, src, [sym, -np.abs(sym), sym], False)
    assert_bits_equal(a, b, "signed zeros under signed taps h=%d" % h)
    allneg = np.full(shape, -0.0, np.float32)
    a, _ = ctx.gauss_hw(allneg, sigma, (h, h, h), None, False)
    assert_bits_equal(a, np.zeros(shape, np.float32), "all -0.0 gives +0.0")


def test_separable_generic_taps(ctx, oracle):
    """ApplySeparable with caller-supplied (non-Gaussian, signed) taps."""
    src = volgen.noise_volume((20, 21, 22), seed=8)
    rng = np.random.default_rng(9)
    taps = [rng.standard_normal(2 * h + 1).astype(np.float32) for h in (3, 2, 4)]
    a, A = ctx.separable3d(src, taps, None, False)
    import ctypes as C
    fp = C.POINTER(C.c_float)
    b = np.empty_like(src)
    oracle.lib.vo_separable3d.restype = C.c_float
    oracle.lib.vo_separable3d.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, fp,
                                          C.c_int, C.c_int]
    B = oracle.lib.vo_separable3d(src.ctypes.data_as(fp), b.ctypes.data_as(fp), None, 22, 21, 20,
                                  taps[0].ctypes.data_as(fp), 3, taps[1].ctypes.data_as(fp), 2,
                                  taps[2].ctypes.data_as(fp), 4, 0)
    assert_bits_equal(a, b, "generic separable")
    assert A == B


# ------------------------------------------------------------------------------------------ blobs
def test_blob_reference_test_command(ctx, oracle):
    """The reference's own test (tests/test_blob_detection.sh:21): 58 scales with a mask; it finds
    11 minima and writes the best one as '235.2 392 313.6 177.915 -140.018'."""
    from visfd_amd import api
    g = golden("blob_rec")
    img = volgen.read_mrc(os.path.join(GOLDEN, "test_blob_detect.rec"))
    msk = volgen.read_mrc(os.path.join(GOLDEN, "test_blob_detect_mask.rec"))
    diam = volgen.cli_blob_diameters(160.0, 280.0, 1.01, 1.0) / np.float32(19.6)
    sig = api.diameters_to_sigmas(diam)
    assert_bits_equal(sig, g["sigmas"], "sigmas")
    mins, maxs = ctx.blob_dog(img, sig, msk, None, 0.02, ratio(oracle), 0.0, -np.inf, False)
    mins = volgen.sort_blobs(mins, True)
    assert len(mins) == 11
    assert_bits_equal(mins, g["minima"], "minima (x,y,z,sigma,score)")
    d = api.sigmas_to_diameters(np.ascontiguousarray(mins[:, 3]))
    w = np.float32(19.6)
    line = "%g %g %g %g %g" % (mins[0, 0] * w, mins[0, 1] * w, mins[0, 2] * w, d[0] * w, mins[0, 4])
    assert line == "235.2 392 313.6 177.915 -140.018"


def test_blob_seeded(ctx, oracle):
    from visfd_amd import api
    g = golden("blob_seeded")
    src = volgen.blob_volume(volgen.BLOB_SHAPE, seed=201)
    mask = volgen.block_mask(volgen.BLOB_SHAPE, seed=202)
    sig = api.diameters_to_sigmas(volgen.BLOB_DIAMS)
    for tag, m in (("nomask", None), ("mask", mask)):
        for mode, kw in volgen.BLOB_MODES.items():
            a, b = ctx.blob_dog(src, sig, m, None, 0.02, ratio(oracle), **kw)
            assert_bits_equal(volgen.sort_blobs(a, True), g["%s_%s_min" % (tag, mode)], tag + mode + " min")
            assert_bits_equal(volgen.sort_blobs(b, False), g["%s_%s_max" % (tag, mode)], tag + mode + " max")


def test_blob_many_candidates(ctx, oracle):
    """Sixty overlapping blobs in noise: dozens of extrema, a stress test of index parity."""
    src = volgen.blob_volume((40, 44, 48), seed=31, nblobs=60)
    sig = np.array([1.2, 1.5, 1.9, 2.4, 3.0, 3.7], np.float32)
    a = ctx.blob_dog(src, sig, None, None, 0.02, 2.5)
    b = oracle.blob_dog(src, sig, None, None, 0.02, 2.5)
    assert len(b[0]) > 30 and len(b[1]) > 30
    assert_bits_equal(volgen.sort_blobs(a[0], True), volgen.sort_blobs(b[0], True), "noise minima")
    assert_bits_equal(volgen.sort_blobs(a[1], False), volgen.sort_blobs(b[1], False), "noise maxima")


def test_blob_pure_noise_fine_scales(ctx, oracle):
    """Pure noise at scales below a voxel: about one voxel in thirty passes the scan's 3x3x3 pre-test (its candidate buffer
    is flushed many times per march); masked and unmasked, rows that span several 64-wide tiles with a ragged last one."""
    shape = (24, 21, 300)
    src = volgen.noise_volume(shape, seed=77)
    mask = volgen.block_mask(shape, seed=78)
    sig = np.array([0.6, 0.7, 0.8], np.float32)
    for m in (None, mask):
        b = oracle.blob_dog(src, sig, m, None, 0.02, 2.5)
        a = ctx.blob_dog(src, sig, m, None, 0.02, 2.5)
        assert len(b[0]) + len(b[1]) > 40
        assert_bits_equal(volgen.sort_blobs(a[0], True), volgen.sort_blobs(b[0], True), "minima")
        assert_bits_equal(volgen.sort_blobs(a[1], False), volgen.sort_blobs(b[1], False), "maxima")


def test_blob_dog_in_two_halves(ctx, oracle):
    """visfd_hip_blob_dog_begin_dev / _end with other work of the same context queued in between (a Gaussian that reuses the
    filter workspaces, a radix select, tensor voting: every workspace the detector shares with other stages): the lists of the
    one-call form, bit for bit; the capacity retry of `end` (the job survives VISFD_HIP_ECAPACITY); abort."""
    import torch
    dev = torch.device("cuda:0")
    shape = (40, 44, 48)
    src_h = volgen.blob_volume(shape, seed=31, nblobs=60)
    sig = np.array([1.2, 1.5, 1.9, 2.4, 3.0, 3.7], np.float32)
    want = oracle.blob_dog(src_h, sig, None, None, 0.02, 2.5)
    src = torch.from_numpy(src_h).to(dev)
    for cap in (1 << 16, 3):
        job = ctx.blob_dog_begin_dev(src, sig, None, None, 0.02, 2.5)
        other = torch.empty_like(src)
        ctx.gauss_dev(src, other, (2.0, 2.0, 2.0), (5, 5, 5))
        sal = (other - other.min()).contiguous()
        ctx.threshold_fraction_dev(sal, 0.1)
        dirs = torch.zeros((3,) + shape, device=dev)
        dirs[0] = 1.0
        ten = torch.empty((6,) + shape, device=dev)
        ctx.tv_dense_stick_dev(sal, dirs, ten, 3.0, 4, 2.0 ** 0.5)
        got = ctx.blob_dog_end(job, cap)
        assert_bits_equal(volgen.sort_blobs(got[0], True), volgen.sort_blobs(want[0], True), "minima, two halves, cap %d" % cap)
        assert_bits_equal(volgen.sort_blobs(got[1], False), volgen.sort_blobs(want[1], False), "maxima, two halves, cap %d" % cap)
        with pytest.raises(ValueError):
            ctx.blob_dog_end(job, cap)
    job = ctx.blob_dog_begin_dev(src, sig, None, None, 0.02, 2.5)
    ctx.blob_dog_abort(job)
    ctx.blob_dog_abort(job)     # (a finished job: nothing to do)
    one = ctx.blob_dog_dev(src, sig, None, None, 0.02, 2.5)
    assert_bits_equal(volgen.sort_blobs(one[0], True), volgen.sort_blobs(want[0], True), "one call after an abort")


def test_blob_tiny_images(ctx, oracle):
    src = volgen.noise_volume((2, 9, 9), seed=3)
    a = ctx.blob_dog(src, np.array([1, 1.3, 1.7], np.float32), None, None, 0.02, 2.5)
    assert len(a[0]) == 0 and len(a[1]) == 0


# ------------------------------------------------------------------------------------------ ridges
def test_hessian_bit_exact(ctx, oracle):
    g = golden("membrane_seeded")
    src = volgen.membrane_volume(volgen.MEM_SHAPE, seed=301)
    for tag, m in (("nomask", None), ("mask", volgen.block_mask(volgen.MEM_SHAPE, seed=302))):
        grad, hess = ctx.calc_hessian(src, volgen.MEM_SIGMA, ratio(oracle), m)
        assert_bits_equal(hess, g[tag + "_hess"], "hessian " + tag)
        assert_bits_equal(grad, g[tag + "_grad"], "gradient " + tag)
    with pytest.raises(Exception):
        ctx.calc_hessian(np.zeros((2, 5, 5), np.float32), 1.0, 2.5)


def test_eigen_solver(ctx, oracle):
    g = golden("eigen")
    mats = g["mats"]
    for oname, order in (("inc", 0), ("dec", 1)):
        d = ctx.diagonalize(mats, order)
        ref = g["diag_" + oname]
        scale = np.max(np.abs(ref[:, :3]), axis=1, keepdims=True) + 1e-30
        err = np.max(np.abs(d[:, :3].astype(np.float64) - ref[:, :3]) / scale)
        assert err <= 1e-5, "eigenvalues rel err %g" % err


def _shoemake_frame(sm):
    """Shoemake triple -> quaternion -> rotation matrix, the decoding half of the reference's frame round trip
    (lin3_utils.hpp:311-337 Shoemake2Quaternion, :275-305 Quaternion2Matrix), in float64 for the checks below."""
    sm = sm.astype(np.float64)
    t1, t2 = 2 * np.pi * sm[:, 1], 2 * np.pi * sm[:, 2]
    r1, r2 = np.sqrt(1.0 - sm[:, 0]), np.sqrt(sm[:, 0])
    q0, q1, q2, q3 = np.sin(t1) * r1, np.cos(t1) * r1, np.sin(t2) * r2, np.cos(t2) * r2
    M = np.empty((len(sm), 3, 3))
    M[:, 0, 0] = 1 - 2 * q2 * q2 - 2 * q3 * q3
    M[:, 1, 1] = 1 - 2 * q1 * q1 - 2 * q3 * q3
    M[:, 2, 2] = 1 - 2 * q1 * q1 - 2 * q2 * q2
    M[:, 0, 1] = 2 * (q1 * q2 - q3 * q0)
    M[:, 1, 0] = 2 * (q1 * q2 + q3 * q0)
    M[:, 1, 2] = 2 * (q2 * q3 - q1 * q0)
    M[:, 2, 1] = 2 * (q2 * q3 + q1 * q0)
    M[:, 0, 2] = 2 * (q1 * q3 + q2 * q0)
    M[:, 2, 0] = 2 * (q1 * q3 - q2 * q0)
    return M


def test_eigen_solver_frames(ctx, oracle):
    """All six outputs of visfd_hip_diagonalize_flat_sym3 (eigen3_simple.hpp:137-342, lin3_utils.hpp:231-375): the three
    eigenvalues AND the Shoemake triple that encodes the eigenvector frame, on tests/golden/eigen.npz -- random matrices
    over six orders of magnitude plus the degenerate, diagonal, tiny and huge cases.  The triple is decoded with the
    reference's formulas and checked three ways: the frame is orthonormal; it diagonalises the matrix
    (V^T diag(lambda) V = M within 1e-5 of the matrix's scale -- this also pins the arbitrary frames of degenerate
    matrices); and where the eigenvalues are well separated every eigenvector equals the reference's up to sign."""
    g = golden("eigen")
    mats = g["mats"]
    full = np.zeros((len(mats), 3, 3), np.float64)
    for (a, b), k in {(0, 0): 0, (1, 1): 1, (2, 2): 2, (0, 1): 3, (1, 2): 4, (0, 2): 5}.items():
        full[:, a, b] = full[:, b, a] = mats[:, k]
    scale = np.max(np.abs(mats), axis=1).astype(np.float64) + 1e-300
    for oname, order in (("inc", 0), ("dec", 1)):
        d = ctx.diagonalize(mats, order)
        ref = g["diag_" + oname]
        assert np.all(np.isfinite(d)), "non-finite output"
        vecs, rvecs = _shoemake_frame(d[:, 3:6]), _shoemake_frame(ref[:, 3:6])     # rows of vecs[i] = eigenvectors
        V = vecs.astype(np.float64)
        eye = np.einsum("nij,nkj->nik", V, V)
        assert np.max(np.abs(eye - np.eye(3))) <= 2e-6, "frame not orthonormal: %g" % np.max(np.abs(eye - np.eye(3)))
        assert np.all(np.linalg.det(V) > 0.99), "frame is not a proper rotation (eigen3_simple.hpp:316-319)"
        recon = np.einsum("nki,nk,nkj->nij", V, d[:, :3].astype(np.float64), V)
        err = np.max(np.abs(recon - full), axis=(1, 2)) / scale
        assert np.max(err) <= 1e-5, "V^T diag(lambda) V != M: rel err %g at case %d" % (np.max(err), int(np.argmax(err)))
        # well-separated spectra: the reference's own eigenvectors, up to sign
        lam = np.sort(ref[:, :3].astype(np.float64), axis=1)
        gap = np.minimum(lam[:, 1] - lam[:, 0], lam[:, 2] - lam[:, 1]) / scale
        sep = gap > 1e-2
        assert sep.sum() > 3000
        dots = np.abs(np.einsum("nij,nij->ni", V[sep], rvecs[sep].astype(np.float64)))
        assert np.min(dots) >= 1.0 - 1e-6, "eigenvector differs from the reference's: |cos| = %.9f" % np.min(dots)
        # exactly degenerate inputs (multiples of the identity, the zero matrix): the reference returns the identity frame
        for i in range(len(mats)):
            if mats[i, 0] == mats[i, 1] == mats[i, 2] and not np.any(mats[i, 3:]):
                assert np.max(np.abs(np.abs(V[i]) - np.eye(3))) <= 1e-6, "frame of a multiple of the identity (case %d)" % i
                assert_close_rel(d[i, :3], ref[i, :3], 1e-6, "eigenvalues of a multiple of the identity")


def test_gauss_512_cubed_crops_equal_oracle(ctx, oracle):
    """BASELINE config 2 at its own size: the separable Gaussian (sigma 2, h = 5) on a synthetic 512^3 volume, single
    sweep == three passes bit for bit, crops at corners / faces / interior equal to the oracle on crop + halo, and the
    tolerance kernel within 1e-5."""
    import torch
    dev = torch.device("cuda:0")
    n = 512
    g = torch.Generator(device=dev).manual_seed(512)
    src = torch.randn((n, n, n), device=dev, generator=g) * 100 + 1000
    sigma, h = (2.0,) * 3, 5
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    ctx.gauss_dev(src, dst, sigma, (h, h, h))
    ctx.synchronize()
    fused = dst.clone()
    torch.cuda.synchronize()
    with ctx.options(gauss_3pass=1):
        ctx.gauss_dev(src, dst, sigma, (h, h, h))
    ctx.synchronize()
    assert torch.equal(fused, dst), "single sweep != three passes at 512^3"
    with ctx.options(gauss_fma=1):
        ctx.gauss_dev(src, dst, sigma, (h, h, h))
    ctx.synchronize()
    assert float((dst - fused).abs().max()) <= 1e-5 * float(fused.abs().max()), "tolerance kernel at 512^3"
    assert not torch.equal(dst, fused)
    got = torch.empty_like(src)
    ctx.gauss_dev(src, got, sigma, (h, h, h), None, False)
    ctx.synchronize()
    E = 24
    for (z0, y0, x0) in [(0, 0, 0), (n - E, n - E, n - E), (250, 0, 488), (100, 300, 200), (n - E, 17, 256)]:
        lo = [max(0, z0 - h), max(0, y0 - h), max(0, x0 - h)]
        hi = [min(n, z0 + E + h), min(n, y0 + E + h), min(n, x0 + E + h)]
        sub = src[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].cpu().numpy().copy()
        want, _ = oracle.gauss_hw(sub, sigma, (h, h, h), None, False)
        a = got[z0:z0 + E, y0:y0 + E, x0:x0 + E].cpu().numpy()
        b = want[z0 - lo[0]:z0 - lo[0] + E, y0 - lo[1]:y0 - lo[1] + E, x0 - lo[2]:x0 - lo[2] + E]
        assert_bits_equal(a, b, "crop at %s" % ((z0, y0, x0),))


def _same_float_field(a, b, what):
    """bitwise equal, except that any NaN equals any NaN (payloads are not part of the contract)"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), "%s: NaN patterns differ (%d vs %d NaNs)" % (what, na.sum(), nb.sum())
    ok = na | (a.view(np.uint32) == b.view(np.uint32))
    assert ok.all(), "%s: %d values differ; first at %s: %r vs %r" % (what, (~ok).sum(), tuple(np.argwhere(~ok)[0]),
                                                                      a[tuple(np.argwhere(~ok)[0])], b[tuple(np.argwhere(~ok)[0])])


@pytest.mark.parametrize("h", [2, 5, 7, 9])
def test_gauss_non_finite_and_sparse_inputs(ctx, oracle, h):
    """NaN / +-Inf voxels and long runs of zeros (the reference's sparse shortcut, filter1d.hpp:59-94, which forces +0.0
    where every sample under the window is zero and otherwise sums NaN/Inf terms like any other): the GPU filters --
    single sweep (h <= 8), three passes (h = 9 and gauss_3pass), masked, DoG/LoG -- reproduce the reference's field, NaN
    for NaN and bit for bit elsewhere."""
    rng = np.random.default_rng(100 + h)
    shape = (30, 37, 52)
    src = (rng.standard_normal(shape) * 100 + 1000).astype(np.float32)
    src[:, :, 20:45] = 0.0                       # a slab of zeros wider than any window
    src[10:25, 5:30, :] *= (rng.random((15, 25, 52)) < 0.1)      # sparse region
    src[3, 4, 5] = np.nan
    src[20, 30, 40] = np.inf                     # inside the zero slab: Inf * tap next to 0 * tap
    src[12, 12, 12] = -np.inf
    src[29, 36, 51] = np.nan                     # a corner
    src[15, 18, 30] = -0.0
    sigma = (h / 2.6,) * 3
    mask = volgen.block_mask(shape, seed=9)
    for opts in ({}, {"gauss_3pass": 1}):
        with ctx.options(**opts):
            for m, norm in ((None, True), (None, False), (mask, True), (mask, False)):
                want, _ = oracle.gauss_hw(src, sigma, (h, h, h), m, norm)
                got, _ = ctx.gauss_hw(src, sigma, (h, h, h), m, norm)
                _same_float_field(got, want, "gaussian h=%d mask=%s normalize=%s %s" % (h, m is not None, norm, opts))
    r = oracle.ratio_from_threshold(0.03)
    s3 = (h / 2.7,) * 3
    _same_float_field(ctx.log(src, s3, 0.02, r)[0], oracle.log(src, s3, 0.02, r)[0], "LoG h=%d" % h)
    _same_float_field(ctx.log(src, s3, 0.02, r, mask)[0], oracle.log(src, s3, 0.02, r, mask)[0], "masked LoG h=%d" % h)


def test_blob_detection_with_non_finite_voxels(ctx, oracle):
    """A NaN voxel poisons the LoG values within a window of it; every comparison with a NaN is false in the reference's
    scan (feature.hpp:245-304), which keeps or drops candidates accordingly: same lists from the GPU."""
    rng = np.random.default_rng(4)
    src = (rng.standard_normal((36, 40, 44)) * 100 + 1000).astype(np.float32)
    src[18, 20, 22] = np.nan
    src[5, 35, 40] = np.inf
    sig = np.array([1.2, 1.5, 1.9, 2.4], np.float32)
    r = oracle.ratio_from_threshold(0.03)
    b = ctx.blob_dog(src, sig, None, None, 0.02, r)
    bo = oracle.blob_dog(src, sig, None, None, 0.02, r)
    for x, y, asc in ((b[0], bo[0], True), (b[1], bo[1], False)):
        fin_x, fin_y = x[np.isfinite(x[:, 4])], y[np.isfinite(y[:, 4])]
        assert_bits_equal(volgen.sort_blobs(fin_x, asc), volgen.sort_blobs(fin_y, asc), "blob lists next to NaN/Inf voxels")
        assert len(x) == len(y), "number of blobs with non-finite scores: %d vs %d" % (len(x) - len(fin_x), len(y) - len(fin_y))


def test_saliency_direction_threshold(ctx, oracle):
    g = golden("membrane_seeded")
    for tag, m in (("nomask", None), ("mask", volgen.block_mask(volgen.MEM_SHAPE, seed=302))):
        hess = g[tag + "_hess"]
        sal, dirs = ctx.hessian_saliency(hess, po.ORDER_DECREASING, m)
        assert_close_rel(sal, g[tag + "_sal"], 1e-5, "saliency " + tag)
        # the principal direction is a unit vector: compare component-wise on that scale, on voxels
        # whose top two eigenvalues are well separated (elsewhere the eigenvector is ill-conditioned)
        ref = g[tag + "_dir"]
        ev = oracle.diagonalize(hess, po.ORDER_DECREASING)
        gap = np.abs(ev[..., 0] - ev[..., 1]) / (np.max(np.abs(ev[..., :3]), axis=-1) + 1e-30)
        ok = gap > 1e-2
        if m is not None:
            ok &= m != 0
        assert ok.mean() > 0.5
        assert np.max(np.abs(dirs[ok] - ref[ok])) <= 2e-5
        # threshold selection is exact given the same saliency input
        s_in = g[tag + "_sal"].copy()
        thr = ctx.threshold_fraction(s_in, volgen.MEM_FRACTION, m)
        assert np.float32(thr) == g[tag + "_thr"]
        assert_bits_equal(s_in, g[tag + "_salthr"], "thresholded saliency " + tag)


# ------------------------------------------------------------------------------------------ tensor voting

def test_blob_scan_overflow_path(ctx, oracle, monkeypatch):
    """The pipelined scale-space scan falls back to the synchronous, buffer-growing scan when a candidate or survivor
    buffer overflows: with buffers of 8 entries (test hook) the lists must still equal the oracle's."""
    src = volgen.blob_volume(volgen.BLOB_SHAPE, seed=201)
    sig = oracle.diameters_to_sigmas(volgen.BLOB_DIAMS)
    r = ratio(oracle)
    want = oracle.blob_dog(src, sig, None, None, 0.02, r, np.inf, -np.inf, False)
    assert len(want[0]) > 8 or len(want[1]) > 8
    for cap in (8, 24, 64):     # every scale overflows / only the scales with the longest lists do
        with ctx.options(blob_test_cap=cap):
            got = ctx.blob_dog(src, sig, None, None, 0.02, r, np.inf, -np.inf, False)
        for g, w, asc in ((got[0], want[0], True), (got[1], want[1], False)):
            assert_bits_equal(volgen.sort_blobs(g, asc), volgen.sort_blobs(w, asc), "blob list through the overflow path, cap %d" % cap)


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_tensor_voting_seeded(ctx, oracle, tag, monkeypatch):
    g = golden("membrane_seeded")
    m = volgen.block_mask(volgen.MEM_SHAPE, seed=302) if tag == "mask" else None
    sal, dirs = g[tag + "_salthr"], g[tag + "_dir"]
    # all three exact kernels (unmasked surfaces with exponent 2 or 4 take the exact form of tv_box.hip unless tv_exact_tiled
    # is set; everything else tv_tiled.hip; tv_dense: the baseline kernel); persistent grids of 1-3 workgroups, so that each
    # claims many units of work; tv_poison: NaN patterns wherever a kernel could read what it has not written
    for opts in ({"tv_dense": 0}, {"tv_dense": 1}, {"tv_max_wg": 3}, {"tv_max_wg": 1, "tv_zrun": 3}, {"tv_exact_tiled": 1},
                 {"tv_exact_tiled": 1, "tv_max_wg": 3}, {"tv_poison": 1, "tv_max_wg": 2}):
        with ctx.options(**opts):
            for ex in (4, 2):
                ten = ctx.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, ex, 2.0 ** 0.5, m, m)
                assert_bits_equal(ten, g["%s_tensor_e%d" % (tag, ex)], "tensor e%d %s" % (ex, opts))
            ten = ctx.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, 3, 2.0 ** 0.5, m, m)
            assert_close_rel(ten, g[tag + "_tensor_e3"], 1e-5, "tensor e3")
            ten = ctx.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, 4, 2.0 ** 0.5, m, m, curves=True)
            assert_bits_equal(ten, g[tag + "_tensor_curves"], "curve-mode tensor")
    ten = g[tag + "_tensor_e4"]
    s2 = sal.copy()
    ctx.tensor_saliency(ten, po.ORDER_DECREASING, s2, m)
    assert_close_rel(s2, g[tag + "_tvsal"], 1e-5, "post-TV saliency")


def test_membrane_reference_fixture_end_to_end(ctx, oracle):
    """tests/test_image_membrane.rec through Hessian -> saliency -> threshold -> TV -> score, every
    stage fed by the previous DEVICE stage, compared with the reference's outputs."""
    g = golden("membrane_rec")
    mem = volgen.read_mrc(os.path.join(GOLDEN, "test_image_membrane.rec"))
    sigma = np.float32(1.5)
    grad, hess = ctx.calc_hessian(mem, sigma, ratio(oracle))
    assert_bits_equal(hess, g["hess"], "hessian")
    sal, dirs = ctx.hessian_saliency(hess, po.ORDER_DECREASING)
    assert_close_rel(sal, g["sal_dec"], 1e-5, "saliency")
    thr = ctx.threshold_fraction(sal, 0.1)
    assert abs(thr - float(g["thr_dec"])) <= 1e-5 * abs(float(g["thr_dec"]))
    kept, kept_ref = sal != 0, g["salthr_dec"] != 0
    # the selected set may differ only where the saliency ties the threshold to within tolerance
    diff = kept != kept_ref
    if diff.any():
        near = np.abs(g["sal_dec"][diff] - g["thr_dec"]) <= 1e-5 * abs(float(g["thr_dec"]))
        assert near.all(), "threshold membership differs away from the tie band"
    ten = ctx.tv_dense_stick(sal, dirs, 4 * sigma / 2, 4, 2.0 ** 0.5)
    assert_close_rel(ten, g["tensor_dec"], 1e-4 if diff.any() else 1e-5, "vote tensor")
    s2 = sal.copy()
    ctx.tensor_saliency(ten, po.ORDER_DECREASING, s2)
    assert_close_rel(s2, g["tvsal_dec"], 1e-4 if diff.any() else 1e-5, "post-TV saliency")


@pytest.mark.parametrize("masked", [False, True])
def test_membrane_peak_height_factor_vs_oracle(ctx, oracle, masked):
    """`-membrane-background`: both score loops multiply by (image - background), background = ApplyGauss(image, sigma_b)
    (bin/filter_mrc/handlers.cpp:1577-1605, :1698-1702, :1883-1887).  The whole stage through the C ABI
    (visfd_hip_membrane_detect_bg) against the oracle's stages chained the same way; a background of width 0 is the
    plain stage."""
    shape = volgen.MEM_SHAPE
    src = volgen.membrane_volume(shape, seed=301)
    m = volgen.block_mask(shape, seed=302) if masked else None
    r = ratio(oracle)
    sigma, sigma_b, frac, sigma_tv = volgen.MEM_SIGMA, np.float32(3.0), volgen.MEM_FRACTION, volgen.MEM_TV_SIGMA
    hb = int(np.floor(np.float32(sigma_b) * np.float32(r)))
    bg, _ = oracle.gauss_hw(src, (sigma_b,) * 3, (hb,) * 3, m, True)
    peak = (src - bg).astype(np.float32)
    _, hess = oracle.calc_hessian(src, sigma, r, m)
    sal_o, dirs_o = oracle.hessian_saliency(hess, po.ORDER_DECREASING, m)
    sal_o = (sal_o * peak).astype(np.float32)
    if m is not None:
        sal_o[m == 0] = 0.0
    raw_o = sal_o.copy()
    thr_o = oracle.threshold_fraction(sal_o, frac, m)
    for sigma_tv_run in (0.0, sigma_tv):
        sal, ten, dirs, thr = ctx.membrane_detect(src, sigma, r, po.ORDER_DECREASING, frac, 0.0, sigma_tv_run, 4, 2.0 ** 0.5, m,
                                                  want_tensor=True, want_dir=True, sigma_background=sigma_b)
        scale = float(np.abs(raw_o).max())
        assert abs(thr - thr_o) <= 1e-5 * scale
        if sigma_tv_run == 0.0:
            kept, kept_o = sal != 0, sal_o != 0
            diff = kept != kept_o
            assert np.all(np.abs(raw_o[diff] - thr_o) <= 1e-5 * scale), "threshold membership differs away from the tie band"
            both_kept = kept & kept_o
            assert np.all(np.abs(sal[both_kept] - sal_o[both_kept]) <= 1e-5 * scale)
            assert (raw_o < 0).any(), "the test volume should have voxels with a negative peak height (dark membranes)"
        else:
            ten_o = oracle.tv_dense_stick(sal_o, dirs_o, sigma_tv, 4, 2.0 ** 0.5, m, m)
            assert_close_rel(ten, ten_o, 1e-4, "vote tensor from peak-height-weighted saliencies")
            s2 = sal_o.copy()
            oracle.tensor_saliency(ten_o, po.ORDER_DECREASING, s2, m)
            want = s2.copy()
            sel = np.ones(shape, bool) if m is None else (m != 0)
            want[sel] = (s2[sel] * peak[sel]).astype(np.float32)
            assert_close_rel(sal, want, 1e-4, "post-vote score times peak height")
    # width 0: the plain stage, bit for bit the same call without the factor
    a = ctx.membrane_detect(src, sigma, r, po.ORDER_DECREASING, frac, 0.0, sigma_tv, 4, 2.0 ** 0.5, m, sigma_background=0.0)
    b = ctx.membrane_detect(src, sigma, r, po.ORDER_DECREASING, frac, 0.0, sigma_tv, 4, 2.0 ** 0.5, m)
    assert_bits_equal(a[0], b[0], "no background == plain stage")


def _sparse_field(shape, seed, fraction=0.06):
    """Random saliency (a fraction non-zero, clustered on a plane) + random unit directions."""
    rng = np.random.default_rng(seed)
    sal = np.zeros(shape, np.float32)
    pick = rng.random(shape) < fraction
    zz = np.arange(shape[0])[:, None, None] + 0 * np.arange(shape[1])[None, :, None]
    pick |= (np.abs(zz - shape[0] // 2) <= 1) & (rng.random(shape) < 0.5)
    sal[pick] = rng.uniform(1.0, 1e6, int(pick.sum())).astype(np.float32)
    d = rng.standard_normal(shape + (3,)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    return sal, np.ascontiguousarray(d, np.float32)


@pytest.mark.parametrize("sigma_tv,shape", [(8.66, (20, 37, 45)), (11.0, (12, 40, 50)), (1.0, (9, 20, 33))])
def test_tensor_voting_wide_windows(ctx, oracle, sigma_tv, shape, monkeypatch):
    """BASELINE config 4's window (sigma_tv = 8.66 -> h = 12) and a window whose sender region is
    split into two bands (h = 15), on ragged tiles, tiled kernel and baseline kernel."""
    sal, dirs = _sparse_field(shape, seed=int(sigma_tv * 10))
    mask = volgen.block_mask(shape, seed=3)
    ref = oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5)
    ref_m = oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5, mask, mask)
    assert np.abs(ref).max() > 0
    # tiled kernel with the chip-filling grid, with 3 and with 1 persistent workgroup (every workgroup then claims
    # many units: ring re-use, per-unit resets), with short runs on top of that, and the baseline kernel
    ref_d = oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5, None, mask)
    for opts in ({}, {"tv_max_wg": 3}, {"tv_max_wg": 1}, {"tv_max_wg": 2, "tv_zrun": 4}, {"tv_max_wg": 2, "tv_no_replay": 1},
                 {"tv_exact_tiled": 1}, {"tv_exact_tiled": 1, "tv_max_wg": 2, "tv_zrun": 4}, {"tv_poison": 1, "tv_max_wg": 2, "tv_zrun": 3},
                 {"tv_dense": 1}):
        if opts.get("tv_dense") and sigma_tv > 9:
            continue
        with ctx.options(**opts):
            ten = ctx.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5)
            assert_bits_equal(ten, ref, "tensor sigma_tv=%g %s" % (sigma_tv, opts))
            ten = ctx.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5, mask, mask)
            assert_bits_equal(ten, ref_m, "masked tensor sigma_tv=%g %s" % (sigma_tv, opts))
            # a destination mask alone (receivers skipped, every sender votes): the exact form of tv_box.hip takes it
            ten = ctx.tv_dense_stick(sal, dirs, sigma_tv, 2, 2.0 ** 0.5, None, mask)
            if opts == {}:
                ref_d2 = oracle.tv_dense_stick(sal, dirs, sigma_tv, 2, 2.0 ** 0.5, None, mask)
            sel = mask != 0
            assert_bits_equal(ten[sel], ref_d2[sel], "destination-masked tensor sigma_tv=%g %s" % (sigma_tv, opts))
    assert np.abs(ref_d).max() > 0
    # a WEIGHTED source mask (values other than 0 and 1 are factors of the votes): the exact form of tv_box.hip declines it
    # after its count pass, tv_tiled.hip takes over
    wmask = (mask * np.random.default_rng(5).choice([0.5, 1.0, 2.0], size=shape)).astype(np.float32)
    assert_bits_equal(ctx.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5, wmask, wmask),
                      oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5, wmask, wmask), "weighted-mask tensor sigma_tv=%g" % sigma_tv)


@pytest.mark.parametrize("sigma_tv,h", [(19.2, 27), (24.1, 34)])
def test_tensor_voting_very_wide_windows(ctx, oracle, sigma_tv, h):
    """Windows far wider than BASELINE's: h = 27 still runs the tiled kernel (two table slices of 48 KB each, one
    workgroup per CU), h = 34 no longer fits its slices into LDS and goes to the baseline kernel."""
    import math
    assert int(math.floor(np.float32(sigma_tv) * np.float32(math.sqrt(2.0)))) == h
    shape = (9, 37, 30)
    sal, dirs = _sparse_field(shape, seed=h)
    ref = oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5)
    assert np.abs(ref).max() > 0
    for opts in ({}, {"tv_max_wg": 2, "tv_zrun": 3}, {"tv_exact_tiled": 1}):
        with ctx.options(**opts):
            assert_bits_equal(ctx.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5), ref, "tensor h=%d %s" % (h, opts))


@pytest.mark.parametrize("nz", [1, 2, 3, 5])
def test_tensor_voting_thin_volumes(ctx, oracle, nz):
    """Volumes of one to five planes (fewer than a pass of the box kernels takes, fewer than the window's reach): both exact
    kernels bit for bit, the tolerance kernel within its contract; odd row and column counts."""
    shape = (nz, 21, 35)
    sal, dirs = _sparse_field(shape, seed=700 + nz, fraction=0.2)
    want = oracle.tv_dense_stick(sal, dirs, 3.0, 4, 2.0 ** 0.5)
    assert np.abs(want).max() > 0
    for opts in ({}, {"tv_exact_tiled": 1}, {"tv_poison": 1, "tv_zrun": 1}):
        with ctx.options(**opts):
            assert_bits_equal(ctx.tv_dense_stick(sal, dirs, 3.0, 4, 2.0 ** 0.5), want, "thin-volume tensor nz=%d %s" % (nz, opts))
    with ctx.options(tv_fma=1, tv_poison=1):
        assert_close_rel(ctx.tv_dense_stick(sal, dirs, 3.0, 4, 2.0 ** 0.5), want, 1e-5, "thin-volume tensor, tolerance mode nz=%d" % nz)


@pytest.mark.parametrize("shape,sigma_tv", [((6, 5, 7), 3.0), ((4, 40, 9), 8.66), ((3, 3, 50), 8.66), ((10, 33, 17), 1.0)])
def test_tensor_voting_narrow_volumes(ctx, oracle, shape, sigma_tv):
    """Volumes narrower than a tile (16 x 32) and than the window: every sender list is shorter than a row of the tile."""
    sal, dirs = _sparse_field(shape, seed=sum(shape), fraction=0.3)
    mask = (np.random.default_rng(5).random(shape) > 0.3).astype(np.float32)
    for m in (None, mask):
        want = oracle.tv_dense_stick(sal, dirs, sigma_tv, 2, 2.0 ** 0.5, m, m)
        for opts in ({}, {"tv_exact_tiled": 1}, {"tv_poison": 1}):
            with ctx.options(**opts):
                assert_bits_equal(ctx.tv_dense_stick(sal, dirs, sigma_tv, 2, 2.0 ** 0.5, m, m), want,
                                  "narrow-volume tensor %s %s mask=%s" % (shape, opts, m is not None))
        with ctx.options(tv_fma=1, tv_poison=1):
            assert_close_rel(ctx.tv_dense_stick(sal, dirs, sigma_tv, 2, 2.0 ** 0.5, m, m), want, 1e-5,
                             "narrow-volume tensor, tolerance mode %s" % (shape,))


def test_tensor_voting_dense_saliency(ctx, oracle):
    """Every voxel salient: the per-band list overflows one 64-entry chunk many times over."""
    shape = (7, 24, 28)
    rng = np.random.default_rng(12)
    sal = rng.uniform(0.5, 2.0, shape).astype(np.float32)
    d = rng.standard_normal(shape + (3,)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    d = np.ascontiguousarray(d, np.float32)
    want = oracle.tv_dense_stick(sal, d, 3.0, 2, 2.0 ** 0.5)
    for opts in ({}, {"tv_exact_tiled": 1}, {"tv_poison": 1}):
        with ctx.options(**opts):
            assert_bits_equal(ctx.tv_dense_stick(sal, d, 3.0, 2, 2.0 ** 0.5), want, "dense-saliency tensor %s" % opts)



@pytest.mark.parametrize("n", [5, 7, 8, 315, 4096, 70001])
def test_threshold_fraction_sizes_and_alignment(ctx, oracle, n):
    """The radix select's histogram reads four values per load where the volume is 16-byte aligned and one by one elsewhere
    (tails, unaligned device pointers, masks): the selected threshold and the thresholded field equal the oracle's for sizes
    around those boundaries, aligned and unaligned, masked and unmasked."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n)
    vals = (rng.standard_normal(n) ** 2).astype(np.float32)
    vals[rng.random(n) < 0.1] = 0.0
    mask = (rng.random(n) > 0.2).astype(np.float32)
    for m in (None, mask):
        want = vals.reshape(1, 1, n).copy()
        thr_o = oracle.threshold_fraction(want, 0.25, None if m is None else m.reshape(1, 1, n))
        for offset in (0, 1):   # a device pointer 4 bytes past a 16-byte boundary
            buf = torch.zeros(n + 4, device=dev)
            sal = buf[offset:offset + n].view(1, 1, n)
            sal.copy_(torch.from_numpy(vals).to(dev).view(1, 1, n))
            dm = None if m is None else torch.from_numpy(m).to(dev).view(1, 1, n)
            thr = ctx.threshold_fraction_dev(sal, 0.25, dm)
            ctx.synchronize()
            assert thr == thr_o, (n, offset, m is not None, thr, thr_o)
            assert_bits_equal(sal.cpu().numpy(), want, "thresholded field n=%d offset=%d" % (n, offset))


def test_ridge_two_step_equals_fused(ctx, oracle):
    """Scores for every voxel + directions of the thresholded survivors (what the pipeline runs) against the fused
    kernel that computes both everywhere: identical scores, identical directions where the score survives, nothing
    written elsewhere; masked and unmasked, both eigenvalue orders, ragged sizes."""
    import torch
    from visfd_amd import api
    dev = torch.device("cuda:0")
    for shape, masked, order in (((37, 45, 70), False, api.DECREASING_EIVALS), ((20, 33, 41), True, api.INCREASING_EIVALS)):
        src = torch.from_numpy(volgen.membrane_volume(shape, seed=91)).to(dev)
        mask = torch.from_numpy(volgen.block_mask(shape, seed=92)).to(dev) if masked else None
        r = api.ratio_from_threshold(0.03)
        sal1 = torch.empty_like(src); dirs1 = torch.zeros((3,) + shape, device=dev)
        ctx.ridge_saliency_dev(src, sal1, dirs1, 1.5, r, order, mask)
        sal2 = torch.empty_like(src); smoothed = torch.empty_like(src)
        dirs2 = torch.full((3,) + shape, 7.0, device=dev)
        ctx.ridge_scores_dev(src, sal2, smoothed, 1.5, r, order, mask)
        ctx.synchronize()
        assert torch.equal(sal1.view(torch.int32), sal2.view(torch.int32)), "scores"
        thr = ctx.threshold_fraction_dev(sal2, 0.1, mask)
        ctx.ridge_directions_dev(smoothed, sal2, dirs2, 1.5, order)
        ctx.synchronize()
        assert thr > 0
        keep = (sal2 != 0)
        assert 0 < int(keep.sum()) < keep.numel()
        for c in range(3):
            assert torch.equal(dirs2[c][keep].view(torch.int32), dirs1[c][keep].view(torch.int32)), "directions"
            assert bool((dirs2[c][~keep] == 7.0).all()), "voxels below the threshold must stay untouched"


def test_tensor_voting_unit_shapes_agree(ctx, oracle, monkeypatch):
    """The voting kernel's units of work (runs of receiver planes with replayed sender planes) must not show in the
    result: runs of 1, 5 and 32 planes, and the path without scratch rings, give the same bits as the CPU
    restatement -- unmasked and masked, with a window (h = 4) much shorter than the volume."""
    shape = (45, 40, 52)
    sal, dirs = _sparse_field(shape, seed=81)
    mask = volgen.block_mask(shape, seed=82)
    want = oracle.tv_dense_stick(sal, dirs, 3.0, 4, 2.0 ** 0.5)
    want_m = oracle.tv_dense_stick(sal, dirs, 3.0, 4, 2.0 ** 0.5, mask, mask)
    for opts in ({}, {"tv_zrun": 1}, {"tv_zrun": 5}, {"tv_no_replay": 1}, {"tv_max_wg": 3}, {"tv_max_wg": 3, "tv_zrun": 1},
                 {"tv_max_wg": 2, "tv_zrun": 5}, {"tv_max_wg": 1, "tv_no_replay": 1}, {"tv_zrun": 2, "tv_max_wg": 5},
                 {"tv_zrun": 3, "tv_poison": 1}, {"tv_zrun": 6, "tv_max_wg": 2}, {"tv_exact_tiled": 1}, {"tv_exact_tiled": 1, "tv_zrun": 5}):
        with ctx.options(**opts):
            assert_bits_equal(ctx.tv_dense_stick(sal, dirs, 3.0, 4, 2.0 ** 0.5), want, "tensor %s" % opts)
            assert_bits_equal(ctx.tv_dense_stick(sal, dirs, 3.0, 4, 2.0 ** 0.5, mask, mask), want_m, "masked tensor %s" % opts)


# ------------------------------------------------------------------------------------------ BASELINE-size volumes
def test_gauss_large_volume_crops_equal_oracle(ctx, oracle):
    """1024 x 1024 x 256 (the BASELINE plane size): the filter is local, so the result inside a crop equals the
    CPU restatement run on the crop plus a halo of h voxels -- bit for bit, for the single-sweep kernel and for
    the three single-axis kernels, in the interior and at faces/corners of the big volume."""
    import torch
    dev = torch.device("cuda:0")
    nz, ny, nx = 256, 1024, 1024
    g = torch.Generator(device=dev).manual_seed(77)
    src = torch.randn((nz, ny, nx), device=dev, generator=g) * 100 + 1000
    sigma, h = (2.0,) * 3, 5
    outs = {}
    dst = torch.empty_like(src)
    torch.cuda.synchronize()   # the context runs on its own stream: the generated volume must be complete first
    ctx.gauss_dev(src, dst, sigma, (h, h, h))
    ctx.synchronize()      # the library runs on its own stream here: finish before torch copies the result
    outs["fused"] = dst.clone()
    torch.cuda.synchronize()
    with ctx.options(gauss_3pass=1):
        ctx.gauss_dev(src, dst, sigma, (h, h, h))
    outs["3-pass"] = dst
    ctx.synchronize()
    assert torch.equal(outs["fused"], outs["3-pass"])
    E = 24   # crop edge
    got = torch.empty_like(src)
    ctx.gauss_dev(src, got, sigma, (h, h, h), None, False)
    ctx.synchronize()
    for (z0, y0, x0) in [(0, 0, 0), (nz - E, ny - E, nx - E), (100, 500, 1000), (7, 1000, 3), (128, 512, 512)]:
        lo = [max(0, z0 - h), max(0, y0 - h), max(0, x0 - h)]
        hi = [min(nz, z0 + E + h), min(ny, y0 + E + h), min(nx, x0 + E + h)]
        sub = src[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].cpu().numpy().copy()
        # normalise=False: the boundary normaliser of the sub-volume differs from the big volume's, so compare
        # the un-normalised filter (the normaliser itself is covered by the small-volume tests)
        want, _ = oracle.gauss_hw(sub, sigma, (h, h, h), None, False)
        a = got[z0:z0 + E, y0:y0 + E, x0:x0 + E].cpu().numpy()
        b = want[z0 - lo[0]:z0 - lo[0] + E, y0 - lo[1]:y0 - lo[1] + E, x0 - lo[2]:x0 - lo[2] + E]
        assert_bits_equal(a, b, "crop at %s" % ((z0, y0, x0),))


def test_blob_large_volume_crops_equal_oracle(ctx, oracle):
    """Scale-space blob detection on a 1024 x 1024 x 96 volume (BASELINE plane size; 12 scales, sigma 2..4, window
    half-widths 5..10, i.e. single-sweep and three-pass filters, pipelined scans): the blobs inside a crop -- far
    enough from the crop's faces that neither a filter window nor the non-max neighbourhood leaves it -- equal, with
    bit-identical scores, the blobs the CPU restatement finds on the crop alone."""
    import torch
    from visfd_amd import pipeline
    dev = torch.device("cuda:0")
    nz, ny, nx = 96, 1024, 1024
    g = torch.Generator(device=dev).manual_seed(4242)
    src = torch.randn((nz, ny, nx), device=dev, generator=g) * 100 + 1000
    sig = pipeline.cli_blob_sigmas(2.0, 4.0, 1.066)
    r = ratio(oracle)
    torch.cuda.synchronize()
    mins, maxs = ctx.blob_dog_dev(src, sig, None, None, 0.02, r, np.inf, -np.inf, False, cap=1 << 22)
    assert len(mins) > 1000 and len(maxs) > 1000
    halo = int(np.floor(r * float(sig[-1]) * 1.01)) + 2      # widest window + the 3x3x3 neighbourhood
    E = 40
    for (z0, y0, x0) in [(halo, 500, 1000 - E), (nz - halo - 30, 60, 120), (30, 1024 - halo - E, 3 * 64 - 5)]:
        ez = min(E, nz - halo - z0)
        lo = [z0 - halo, y0 - halo, x0 - halo]
        hi = [z0 + ez + halo, y0 + E + halo, x0 + E + halo]
        assert min(lo) >= 0 and hi[0] <= nz and hi[1] <= ny and hi[2] <= nx
        sub = src[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].cpu().numpy().copy()
        wmin, wmax = oracle.blob_dog(sub, sig, None, None, 0.02, r, np.inf, -np.inf, False)
        for got, want, asc in ((mins, wmin, True), (maxs, wmax, False)):
            inside = ((got[:, 0] >= x0) & (got[:, 0] < x0 + E) & (got[:, 1] >= y0) & (got[:, 1] < y0 + E) &
                      (got[:, 2] >= z0) & (got[:, 2] < z0 + ez))
            a = got[inside].copy()
            a[:, 0] -= lo[2]; a[:, 1] -= lo[1]; a[:, 2] -= lo[0]
            keep = ((want[:, 0] >= halo) & (want[:, 0] < halo + E) & (want[:, 1] >= halo) & (want[:, 1] < halo + E) &
                    (want[:, 2] >= halo) & (want[:, 2] < halo + ez))
            b = want[keep]
            assert len(b) > 0
            assert_bits_equal(volgen.sort_blobs(a, asc), volgen.sort_blobs(b, asc), "blobs in crop at %s" % ((z0, y0, x0),))


def test_gauss_2048_cubed_crops_equal_oracle(ctx, oracle):
    """2048^3 float32 (the north-star target size; 2^33 voxels, beyond 32-bit indexing and beyond what the reference
    can allocate): single-sweep kernel == three single-axis kernels on the whole 32 GiB volume, and crops in the
    interior, across the 2^31- and 2^32-voxel marks and at faces equal the CPU restatement run on the crop + halo.
    The normalised result in the interior is the un-normalised one divided by the constant (Dx*Dy)*Dz."""
    import torch
    dev = torch.device("cuda:0")
    free, _total = torch.cuda.mem_get_info()
    n = 2048
    if free < 140 * 2 ** 30:
        pytest.skip("needs ~130 GiB of free HBM (two 32 GiB volumes + two 32 GiB pass workspaces)")
    g = torch.Generator(device=dev).manual_seed(2048)
    src = torch.empty((n, n, n), device=dev, dtype=torch.float32)
    for z in range(0, n, 256):   # generate in slabs: randn's temporaries stay small
        src[z:z + 256] = torch.randn((256, n, n), device=dev, generator=g) * 100 + 1000
    sigma, h = (2.0,) * 3, 5
    got = torch.empty_like(src)
    torch.cuda.synchronize()   # the context runs on its own stream: the generated volume must be complete first
    ctx.gauss_dev(src, got, sigma, (h, h, h), None, False)
    ctx.synchronize()
    E = 16
    corners = [(0, 0, 0), (n - E, n - E, n - E), (511, 2040, 2040), (512, 0, 0), (1023, 2047 - E, 1000),
               (1024, 3, 2030), (1500, 1000, 7)]
    for (z0, y0, x0) in corners:
        lo = [max(0, z0 - h), max(0, y0 - h), max(0, x0 - h)]
        hi = [min(n, z0 + E + h), min(n, y0 + E + h), min(n, x0 + E + h)]
        sub = src[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].cpu().numpy().copy()
        want, _ = oracle.gauss_hw(sub, sigma, (h, h, h), None, False)
        a = got[z0:z0 + E, y0:y0 + E, x0:x0 + E].cpu().numpy()
        b = want[z0 - lo[0]:z0 - lo[0] + E, y0 - lo[1]:y0 - lo[1] + E, x0 - lo[2]:x0 - lo[2] + E]
        assert_bits_equal(a, b, "2048^3 crop at %s" % ((z0, y0, x0),))
    # the three single-axis kernels on the same volume
    with ctx.options(gauss_3pass=1):
        alt = torch.empty_like(src)
        ctx.gauss_dev(src, alt, sigma, (h, h, h), None, False)
        ctx.synchronize()
    for z in range(0, n, 256):
        assert torch.equal(got[z:z + 256], alt[z:z + 256]), "fused != 3-pass in slab %d" % z
    # normalised: interior voxels are divided by one constant (filter3d.hpp:1016-1018)
    ctx.gauss_dev(src, alt, sigma, (h, h, h))
    ctx.synchronize()
    d1 = np.float32(0.0)    # D of an interior line: the 1-D filter applied to ones, summed in tap order
    for t in np.asarray(oracle.gauss_taps(2.0, h), np.float32):
        d1 = np.float32(d1 + np.float32(t * np.float32(1.0)))
    dconst = np.float32(np.float32(d1 * d1) * d1)
    for (z0, y0, x0) in [(511, 1000, 1000), (1024, 300, 1700), (1500, 1000, 7 + h)]:
        a = alt[z0:z0 + E, y0:y0 + E, x0:x0 + E].cpu().numpy()
        b = (got[z0:z0 + E, y0:y0 + E, x0:x0 + E].cpu().numpy() / dconst).astype(np.float32)
        assert_bits_equal(a, b, "2048^3 normalised crop at %s" % ((z0, y0, x0),))


# ------------------------------------------------------------------------------------------ empty / degenerate inputs
def test_empty_and_degenerate_inputs(ctx, oracle):
    """No salient voxel, a single salient voxel, a single-plane volume, a constant image: the cases where lists
    are empty, windows are clipped on every side, or nothing can be an extremum."""
    from visfd_amd import api
    shape = (9, 20, 24)
    d = np.zeros(shape + (3,), np.float32)
    d[..., 2] = 1.0
    zero = np.zeros(shape, np.float32)
    for dense in (0, 1):
        ctx.set_option("tv_dense", dense)
        try:
            ten = ctx.tv_dense_stick(zero, d, 3.0, 4, 2.0 ** 0.5)
            assert not ten.any(), "votes without senders"
            one = zero.copy()
            one[4, 10, 12] = 2.5
            assert_bits_equal(ctx.tv_dense_stick(one, d, 3.0, 4, 2.0 ** 0.5), oracle.tv_dense_stick(one, d, 3.0, 4, 2.0 ** 0.5),
                              "single sender")
            flat = np.random.default_rng(3).uniform(0, 1, (1, 20, 24)).astype(np.float32) * (np.arange(24) % 5 == 0)
            dflat = np.ascontiguousarray(d[:1])
            assert_bits_equal(ctx.tv_dense_stick(flat, dflat, 3.0, 4, 2.0 ** 0.5),
                              oracle.tv_dense_stick(flat, dflat, 3.0, 4, 2.0 ** 0.5), "single-plane volume")
        finally:
            ctx.set_option("tv_dense", 0)
    const = np.full((12, 14, 16), 7.0, np.float32)
    sig = np.array([1.0, 1.3, 1.7, 2.2], np.float32)
    mins, maxs = ctx.blob_dog(const, sig, None, None, 0.02, oracle.ratio_from_threshold(0.03))
    a, b = oracle.blob_dog(const, sig, None, None, 0.02, oracle.ratio_from_threshold(0.03))
    assert len(mins) == len(a) and len(maxs) == len(b)
    with pytest.raises(api.VisfdHipError):
        ctx.threshold_fraction(zero.copy(), 1.5)          # the fraction selects no voxel
    mask0 = np.zeros(shape, np.float32)
    out, _ = ctx.gauss_hw(np.ones(shape, np.float32), (1.0,) * 3, (2, 2, 2), mask0, True)
    want, _ = oracle.gauss_hw(np.ones(shape, np.float32), (1.0,) * 3, (2, 2, 2), mask0, True)
    assert_bits_equal(out, want, "all-masked Gaussian")


def test_tensor_voting_large_volume_crops(ctx, oracle):
    """Voting is local (window half-width h): inside a crop, at least h voxels from the crop's faces, the vote tensor
    of the crop alone equals the tensor of the whole volume -- bit for bit, although tiles, sender lists and flush
    boundaries fall differently (the crop's origin is not a multiple of the tile size).  One crop is also checked
    against the CPU restatement."""
    shape, sigma_tv, h = (96, 208, 272), 8.66, 12
    rng = np.random.default_rng(31)
    sal = (rng.random(shape) < 0.05).astype(np.float32) * rng.uniform(0.5, 3.0, shape).astype(np.float32)
    sal[40:44, 60:140, 50:200] = rng.uniform(1.0, 2.0, (4, 80, 150)).astype(np.float32)     # a dense sheet
    d = rng.standard_normal(shape + (3,)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    d = np.ascontiguousarray(d, np.float32)
    full = ctx.tv_dense_stick(sal, d, sigma_tv, 4, 2.0 ** 0.5)
    for (z0, y0, x0), (cz, cy, cx), check_oracle in (((21, 37, 53), (52, 70, 90), False), ((30, 50, 45), (34, 44, 52), True)):
        s_c = np.ascontiguousarray(sal[z0:z0 + cz, y0:y0 + cy, x0:x0 + cx])
        d_c = np.ascontiguousarray(d[z0:z0 + cz, y0:y0 + cy, x0:x0 + cx])
        part = ctx.tv_dense_stick(s_c, d_c, sigma_tv, 4, 2.0 ** 0.5)
        a = part[h:cz - h, h:cy - h, h:cx - h]
        b = full[z0 + h:z0 + cz - h, y0 + h:y0 + cy - h, x0 + h:x0 + cx - h]
        assert a.size > 0 and np.abs(a).max() > 0
        assert_bits_equal(a, b, "crop at %s" % ((z0, y0, x0),))
        if check_oracle:
            assert_bits_equal(part, oracle.tv_dense_stick(s_c, d_c, sigma_tv, 4, 2.0 ** 0.5), "crop vs oracle")
