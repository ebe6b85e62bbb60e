"""The TOLERANCE modes of the HIP path (context options tv_fma / gauss_fma) against the CPU oracle.

BASELINE.json's north_star asks for bit-exact indices and for float voxel values within 1e-5 relative.  The default
kernels reproduce the reference's bits; the tolerance modes trade those bits for fused multiply-adds where the
output is a float field (vote tensors, a plain Gaussian).  Bar here: |got - want| <= 1e-5 * max|want| on every case
the exact kernels are tested on (same inputs, same option sweeps), and index-valued results downstream unchanged.
Everything that feeds an index comparison (LoG -> non-max scan) ignores the options and stays exact; that is
asserted too."""
import numpy as np
import pytest

import volgen
from conftest import assert_bits_equal, assert_close_rel, golden
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

TOL = 1e-5
# the per-voxel reading of the same tolerance (conftest.assert_close_rel): the fraction of significant voxels (|want| above
# 1e-3 of the field's scale) allowed to differ by more than 1e-5 of THEIR OWN value.  A vote tensor is a sum of hundreds of
# signed terms: its rounding error scales with the sum of their magnitudes, not with the result, so small results of large
# cancellations exceed 1e-5 of themselves while staying ~1e-7 of the field's scale.  Measured: profiles/r04_tolerance_pervoxel.txt
PV = 0.005


@pytest.fixture(scope="module")
def ctx():
    from visfd_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _sparse_field(shape, seed, frac=0.05):
    rng = np.random.default_rng(seed)
    sal = rng.random(shape, dtype=np.float32)
    sal[rng.random(shape) > frac] = 0.0
    d = rng.standard_normal(shape + (3,)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    return sal, np.ascontiguousarray(d)


# ------------------------------------------------------------------------------------ tensor voting
@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_tv_fma_seeded_goldens(ctx, tag):
    g = golden("membrane_seeded")
    m = volgen.block_mask(volgen.MEM_SHAPE, seed=302) if tag == "mask" else None
    sal, dirs = g[tag + "_salthr"], g[tag + "_dir"]
    for opts in ({}, {"tv_max_wg": 3}, {"tv_max_wg": 1, "tv_zrun": 3}, {"tv_no_fold": 1}):
        with ctx.options(tv_fma=1, tv_poison=1, **opts):
            for ex in (4, 2):
                ten = ctx.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, ex, 2.0 ** 0.5, m, m)
                want = g["%s_tensor_e%d" % (tag, ex)]
                assert_close_rel(ten, want, TOL, "fma tensor e%d %s" % (ex, opts), pervoxel=PV)
                assert not np.array_equal(ten.view(np.uint32), want.view(np.uint32)) or not np.any(want), \
                    "the tolerance mode did not run (bits equal the exact kernel's)"
            # exponent 3 and curve mode have no tolerance form: the option must leave them exact / as before
            ten = ctx.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, 4, 2.0 ** 0.5, m, m, curves=True)
            assert_bits_equal(ten, g[tag + "_tensor_curves"], "curve-mode tensor under tv_fma")
    # the post-vote score from the tolerance-mode tensor
    with ctx.options(tv_fma=1, tv_poison=1):
        ten = ctx.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, 4, 2.0 ** 0.5, m, m)
    s2 = sal.copy()
    ctx.tensor_saliency(ten, po.ORDER_DECREASING, s2, m)
    assert_close_rel(s2, g[tag + "_tvsal"], TOL, "post-TV saliency from the tolerance-mode tensor", pervoxel=PV)


@pytest.mark.parametrize("sigma_tv,shape", [(8.66, (20, 37, 45)), (11.0, (12, 40, 50)), (1.0, (9, 20, 33)), (3.0, (41, 33, 70))])
def test_tv_fma_windows(ctx, oracle, sigma_tv, shape):
    sal, dirs = _sparse_field(shape, seed=int(sigma_tv * 10))
    mask = volgen.block_mask(shape, seed=3)
    ref = oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5)
    ref_m = oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5, mask, mask)
    ref2 = oracle.tv_dense_stick(sal, dirs, sigma_tv, 2, 2.0 ** 0.5)
    for opts in ({}, {"tv_max_wg": 3}, {"tv_max_wg": 1}, {"tv_max_wg": 2, "tv_zrun": 4}, {"tv_max_wg": 2, "tv_no_fold": 1},
                 {"tv_zrun": 1}, {"tv_zrun": 5}):
        with ctx.options(tv_fma=1, tv_poison=1, **opts):
            assert_close_rel(ctx.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5), ref, TOL, "fma tensor %g %s" % (sigma_tv, opts), pervoxel=PV)
            assert_close_rel(ctx.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5, mask, mask), ref_m, TOL,
                             "fma masked tensor %g %s" % (sigma_tv, opts), pervoxel=PV)
            assert_close_rel(ctx.tv_dense_stick(sal, dirs, sigma_tv, 2, 2.0 ** 0.5), ref2, TOL, "fma tensor e2 %g %s" % (sigma_tv, opts), pervoxel=PV)


def test_tv_fma_dense_saliency_and_empty(ctx, oracle):
    """every voxel a sender (the longest sums: 7 153 votes per receiver at h = 12 would take the oracle minutes; h = 4 here),
    and the degenerate inputs"""
    rng = np.random.default_rng(5)
    shape = (20, 24, 40)
    sal = (rng.random(shape, dtype=np.float32) + 0.1).astype(np.float32)
    d = rng.standard_normal(shape + (3,)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    with ctx.options(tv_fma=1, tv_poison=1):
        for ex in (2, 4):
            assert_close_rel(ctx.tv_dense_stick(sal, d, 3.0, ex, 2.0 ** 0.5), oracle.tv_dense_stick(sal, d, 3.0, ex, 2.0 ** 0.5), TOL,
                             "dense-saliency tensor e%d" % ex, pervoxel=PV)
        zero = np.zeros((9, 10, 11), np.float32)
        dz = np.zeros((9, 10, 11, 3), np.float32)
        assert not np.any(ctx.tv_dense_stick(zero, dz, 3.0, 4, 2.0 ** 0.5))
        one = zero.copy()
        one[4, 5, 6] = 2.5
        dz[4, 5, 6] = (0.6, 0.0, 0.8)
        assert_close_rel(ctx.tv_dense_stick(one, dz, 3.0, 4, 2.0 ** 0.5), oracle.tv_dense_stick(one, dz, 3.0, 4, 2.0 ** 0.5), TOL,
                         "single sender", pervoxel=PV)


def test_tv_fma_large_magnitudes(ctx, oracle):
    """saliencies of real tomograms span many orders of magnitude ((l0^2 - l1^2)^2 of Hessian eigenvalues): the tolerance is
    relative to the field's scale whatever that scale is"""
    sal, dirs = _sparse_field((18, 30, 40), seed=77)
    for scale in (1e-12, 1.0, 1e12):
        s = (sal * np.float32(scale)).astype(np.float32)
        with ctx.options(tv_fma=1, tv_poison=1):
            assert_close_rel(ctx.tv_dense_stick(s, dirs, 3.0, 4, 2.0 ** 0.5), oracle.tv_dense_stick(s, dirs, 3.0, 4, 2.0 ** 0.5), TOL,
                             "saliency scale %g" % scale, pervoxel=PV)


# ------------------------------------------------------------------------------------------ Gaussian
@pytest.mark.parametrize("shape", [(40, 50, 70), (33, 17, 129), (7, 9, 200), (64, 64, 64)])
@pytest.mark.parametrize("h", [1, 2, 3, 4, 5, 6, 7, 8])
def test_gauss_fma_vs_oracle(ctx, oracle, shape, h):
    rng = np.random.default_rng(h * 100 + shape[0])
    src = (rng.standard_normal(shape) * 100 + 1000).astype(np.float32)
    sigma = (h / 2.6,) * 3
    want, A = oracle.gauss_hw(src, sigma, (h, h, h))
    with ctx.options(gauss_fma=1):
        got, A2 = ctx.gauss_hw(src, sigma, (h, h, h))
    assert A == A2
    assert_close_rel(got, want, TOL, "fma gaussian h=%d %s" % (h, shape), pervoxel=PV)
    # zero-mean data: the relative bar is against the field's own (small) scale
    src0 = rng.standard_normal(shape).astype(np.float32)
    want0, _ = oracle.gauss_hw(src0, sigma, (h, h, h))
    with ctx.options(gauss_fma=1):
        got0, _ = ctx.gauss_hw(src0, sigma, (h, h, h))
    assert_close_rel(got0, want0, TOL, "fma gaussian of zero-mean noise h=%d %s" % (h, shape), pervoxel=PV)


def test_gauss_fma_leaves_index_paths_exact(ctx, oracle):
    """LoG/DoG feed the 4-D non-max scan, whose strict comparisons need the reference's bits: gauss_fma must not touch them"""
    rng = np.random.default_rng(11)
    src = (rng.standard_normal((30, 40, 50)) * 100 + 1000).astype(np.float32)
    r = oracle.ratio_from_threshold(0.03)
    want = oracle.log(src, (2.0, 2.0, 2.0), 0.02, r)[0]
    sig = np.array([1.5, 1.9, 2.4, 3.0], np.float32)
    bo = oracle.blob_dog(src, sig, None, None, 0.02, r)
    with ctx.options(gauss_fma=1, tv_fma=1):
        got = ctx.log(src, (2.0, 2.0, 2.0), 0.02, r)[0]
        b = ctx.blob_dog(src, sig, None, None, 0.02, r)
    assert_bits_equal(got, want, "LoG under gauss_fma")
    for x, y, asc in ((b[0], bo[0], True), (b[1], bo[1], False)):
        assert_bits_equal(volgen.sort_blobs(x, asc), volgen.sort_blobs(y, asc), "blob list under gauss_fma")


# ------------------------------------------------------------------- vote weight sums (normalisation denominators)
@pytest.mark.parametrize("sigma_tv,shape", [(3.0, (17, 21, 26)), (8.66, (12, 30, 33)), (30.0, (6, 9, 11))])
def test_tv_weight_sum_kernels_agree(ctx, oracle, sigma_tv, shape):
    """visfd_hip_tv_weight_sum (the denominators of TVDenseStick(normalize = true), feature.hpp:1761-1822, 2376-2382): the
    tiled kernel, the baseline kernel's weights-only form (option tv_dense, and the fallback for windows the tiled kernel
    declines: h = 42 here) and a direct numpy sum in the reference's order agree bit for bit."""
    sal, _ = _sparse_field(shape, seed=int(sigma_tv * 7), frac=0.2)
    mask = volgen.block_mask(shape, seed=5)
    h, w, _ = oracle.tv_tables(sigma_tv, 2.0 ** 0.5)
    w = np.asarray(w, np.float32).reshape(2 * h + 1, 2 * h + 1, 2 * h + 1)
    nz, ny, nx = shape
    for m in (None, mask):
        want = np.zeros(shape, np.float32)
        for iz in range(nz):
            for iy in range(ny):
                for ix in range(nx):
                    if m is not None and m[iz, iy, ix] == 0:
                        continue
                    acc = np.float32(0)
                    for jz in range(max(-h, iz - nz + 1), min(h, iz) + 1):       # sender = receiver - j, in bounds
                        for jy in range(max(-h, iy - ny + 1), min(h, iy) + 1):
                            row_s = sal[iz - jz, iy - jy]
                            row_m = m[iz - jz, iy - jy] if m is not None else None
                            for jx in range(max(-h, ix - nx + 1), min(h, ix) + 1):
                                fv = w[jz + h, jy + h, jx + h]
                                if row_m is not None:
                                    if row_m[ix - jx] == 0:
                                        continue
                                    fv = np.float32(fv * row_m[ix - jx])
                                if row_s[ix - jx] == 0 or fv == 0:
                                    continue
                                acc = np.float32(acc + fv)
                    want[iz, iy, ix] = acc
        got = {}
        for dense in (0, 1):
            with ctx.options(tv_dense=dense):
                got[dense] = ctx.tv_weight_sum(sal, sigma_tv, 2.0 ** 0.5, m, m)
        assert_bits_equal(got[0], got[1], "weight sums: tiled (or its fallback) vs baseline kernel, sigma_tv=%g mask=%s" % (sigma_tv, m is not None))
        assert_bits_equal(got[1], want, "weight sums vs the direct sum, sigma_tv=%g mask=%s" % (sigma_tv, m is not None))


@pytest.mark.parametrize("sigma_tv,h", [(19.2, 27), (24.1, 34)])
def test_tv_fma_very_wide_windows(ctx, oracle, sigma_tv, h):
    """windows whose table slices fill most of a CU's LDS (one workgroup per CU) and whose regions need the one-plane lister"""
    shape = (7, 40, 21)
    sal, dirs = _sparse_field(shape, seed=h, frac=0.02)
    ref = oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5)
    for opts in ({}, {"tv_max_wg": 2, "tv_zrun": 3}):
        with ctx.options(tv_fma=1, tv_poison=1, **opts):
            assert_close_rel(ctx.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5), ref, TOL, "fma tensor h=%d %s" % (h, opts), pervoxel=PV)


def test_tv_fma_receiver_plane_ranges(ctx, oracle):
    """the slab form (visfd_hip_tv_dense_stick_slab_dev: receiver planes [z0, z1) of a local array, as the multi-GPU host
    votes its interior and its two bands) in tolerance mode: odd and even ranges, ranges of one plane, the whole array in
    three pieces -- each piece within 1e-5 of the oracle's planes and untouched outside its range"""
    import torch
    shape = (23, 26, 35)
    sal, dirs = _sparse_field(shape, seed=99, frac=0.08)
    want = oracle.tv_dense_stick(sal, dirs, 3.0, 4, 2.0 ** 0.5)
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        from visfd_amd import api
        c2 = api.Context(0, st.cuda_stream)
        tsal = torch.from_numpy(sal).to(dev)
        tdir = torch.from_numpy(np.ascontiguousarray(np.moveaxis(dirs, -1, 0))).to(dev)
        scale = float(np.abs(want).max())
        for ranges in ([(0, 23)], [(0, 7), (7, 18), (18, 23)], [(4, 5), (5, 16)], [(3, 4)], [(0, 1), (22, 23)]):
            ten = torch.full((6,) + shape, 7.5, device=dev)
            with c2.options(tv_fma=1):
                for (a, b) in ranges:
                    c2.tv_dense_stick_dev(tsal, tdir, ten, 3.0, 4, 2.0 ** 0.5, None, None, False, (a, b))
            c2.synchronize()
            got = np.moveaxis(ten.cpu().numpy(), 0, -1)
            covered = np.zeros(shape[0], bool)
            for (a, b) in ranges:
                covered[a:b] = True
                assert np.max(np.abs(got[a:b] - want[a:b])) <= TOL * scale, "planes %d..%d" % (a, b)
            assert np.all(got[~covered] == 7.5), "planes outside the requested ranges were written"
        c2.close()


def test_tv_fma_negative_saliencies(ctx, oracle):
    """Senders with a NEGATIVE saliency (never so on the membrane path -- the score is a sum of squares -- but the entry point
    takes any field): mixed signs, all negative, and a single negative sender."""
    shape = (14, 40, 37)
    sal, dirs = _sparse_field(shape, seed=77, frac=0.08)
    rng = np.random.default_rng(5)
    mixed = sal * np.where(rng.random(shape) < 0.4, -1.0, 1.0).astype(np.float32)
    one = sal.copy()
    zz, yy, xx = np.nonzero(one)
    one[zz[3], yy[3], xx[3]] *= -1.0
    for field, what in ((mixed, "mixed signs"), (-sal, "all negative"), (one, "one negative sender")):
        for ex in (4, 2):
            ref = oracle.tv_dense_stick(field, dirs, 3.0, ex, 2.0 ** 0.5)
            with ctx.options(tv_fma=1, tv_poison=1):
                got = ctx.tv_dense_stick(field, dirs, 3.0, ex, 2.0 ** 0.5)
            assert_close_rel(got, ref, TOL, "fma tensor, %s, exponent %d" % (what, ex), pervoxel=PV)
            with ctx.options(tv_fma=1, tv_max_wg=2, tv_zrun=3):
                got = ctx.tv_dense_stick(field, dirs, 3.0, ex, 2.0 ** 0.5)
            assert_close_rel(got, ref, TOL, "fma tensor, %s, exponent %d, few workgroups" % (what, ex), pervoxel=PV)
