"""BASELINE configs 3 and 4 at FULL size (1024^3, the bench's own synthetic volume and parameters), checked against the
CPU restatement through locality: every stage of the path has a finite reach, so the result inside a crop equals the
oracle's result on the crop plus a halo.  The crops sit at corners and faces of the volume, in its interior, across
the seams of the voting kernel's units of work (tiles of 8 x 32 voxels, runs of 32 receiver planes) and on the
synthetic membranes, where sender lists are long.  At this size every persistent workgroup of the voting kernel claims
more than a hundred units, which no small test reaches.

Tensor voting is compared bit for bit: the oracle votes on the DEVICE's thresholded saliency and directions of the
crop + halo (handlers.cpp:1751-1835 feeds TVDenseStick the same way).  Eigen-derived fields (saliency, post-vote
score) are compared within 1e-5 of the field's scale, blob lists bit for bit."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import volgen  # noqa: E402
from conftest import assert_bits_equal, assert_close_rel  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

pytestmark = pytest.mark.gpu

N = 1024


@pytest.fixture(scope="module")
def gpu():
    import torch
    from visfd_amd import api
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = api.Context(0, stream.cuda_stream)
    yield torch, dev, ctx
    ctx.close()


@pytest.fixture(scope="module")
def volume(gpu):
    """The volume bench.py times (same generator, same seed)."""
    import bench
    torch, dev, ctx = gpu
    src = bench.synth_volume(torch, ctx, (N, N, N), dev, seed=12345)
    torch.cuda.synchronize()
    return src


def _crop(t, lo, hi):
    return t[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].contiguous().cpu().numpy()


def _check_membrane(gpu, oracle, src, crops, E, dense=0.12, ridge_crops=4, tv_fma=0):
    """The steps of pipeline.membrane_detect on `src` (every stage fed by the previous device stage, as in bench.py),
    keeping the thresholded saliency the votes are cast from; then crops against the oracle on crop + halo."""
    import bench
    from visfd_amd import api
    torch, dev, ctx = gpu
    shape = tuple(src.shape)
    dims = np.array(shape)
    nvox = int(np.prod(dims))
    P = bench.MEMBRANE
    sigma = P["sigma"]
    ratio = api.ratio_from_threshold(0.03)
    order = api.DECREASING_EIVALS
    sigma_tv = float(np.float32(P["tv_sigma_ratio"]) * np.float32(sigma))
    h = int(math.floor(np.float32(sigma_tv) * np.float32(math.sqrt(2.0))))
    assert h == 12
    sal = torch.empty_like(src)
    smoothed = torch.empty_like(src)
    dirs = torch.zeros((3,) + shape, device=dev)
    ten = torch.empty((6,) + shape, device=dev)
    ctx.ridge_scores_dev(src, sal, smoothed, sigma, ratio, order)
    thr = ctx.threshold_fraction_dev(sal, P["best_fraction"])
    ctx.ridge_directions_dev(smoothed, sal, dirs, sigma, order)
    ctx.synchronize()
    del smoothed
    sal_thr = sal.clone()
    with ctx.options(tv_fma=tv_fma):
        ctx.tv_dense_stick_dev(sal, dirs, ten, sigma_tv, P["tv_exponent"], math.sqrt(2.0))
    ctx.tensor_saliency_dev(ten, sal, order)
    ctx.synchronize()
    nsal = int((sal_thr != 0).sum().item())
    assert abs(nsal / nvox - P["best_fraction"]) < 1e-3 and thr > 0

    seen_dense = False
    for (z0, y0, x0) in crops:
        c0 = np.array([z0, y0, x0])
        lo = np.maximum(c0 - h, 0)
        hi = np.minimum(c0 + E + h, dims)
        s_sub = _crop(sal_thr, lo, hi)
        d_sub = np.ascontiguousarray(np.stack([_crop(dirs[k], lo, hi) for k in range(3)], axis=-1))
        want = oracle.tv_dense_stick(s_sub, d_sub, sigma_tv, P["tv_exponent"], 2.0 ** 0.5)
        o = c0 - lo
        want_c = want[o[0]:o[0] + E, o[1]:o[1] + E, o[2]:o[2] + E]
        got_c = np.ascontiguousarray(np.stack([_crop(ten[k], c0, c0 + E) for k in range(6)], axis=-1))
        assert np.abs(want_c).max() > 0
        if tv_fma:   # tolerance mode: 1e-5 of the crop's own scale (stricter than the field's)
            assert_close_rel(got_c, want_c, 1e-5, "tolerance-mode vote tensor in the crop at %s" % ((z0, y0, x0),), pervoxel=0.02)
            assert not np.array_equal(got_c.view(np.uint32), want_c.view(np.uint32)), "the tolerance kernel did not run"
        else:
            assert_bits_equal(got_c, want_c, "vote tensor in the crop at %s" % ((z0, y0, x0),))
        seen_dense = seen_dense or (s_sub != 0).mean() > dense
        # post-vote score of the crop (eigen-derived: 1e-5 of the field's scale)
        s2 = np.zeros(want_c.shape[:3], np.float32)
        oracle.tensor_saliency(np.ascontiguousarray(want_c), po.ORDER_DECREASING, s2)
        assert_close_rel(_crop(sal, c0, c0 + E), s2, 1e-5, "post-vote score in the crop at %s" % ((z0, y0, x0),))
    assert seen_dense, "no crop sits on a membrane (sender density well above the 5 % average)"

    # the ridge stage that fed the votes, at full size: saliency before thresholding within 1e-5 of the oracle's on a
    # crop + halo (Gaussian window + finite differences), and the threshold cut consistent with it
    hg = int(math.floor(np.float32(sigma) * np.float32(ratio))) + 1
    for (z0, y0, x0) in crops[:ridge_crops]:
        c0 = np.array([z0, y0, x0])
        lo = np.maximum(c0 - hg, 0)
        hi = np.minimum(c0 + E + hg, dims)
        sub = _crop(src, lo, hi)
        _, hess = oracle.calc_hessian(sub, np.float32(sigma), ratio)
        s_o, _ = oracle.hessian_saliency(hess, po.ORDER_DECREASING)
        o = c0 - lo
        # faces of the crop that are interior to the big volume see a different boundary normaliser: compare away from them
        inner = [slice(o[k] + (hg if lo[k] > 0 else 0), o[k] + E - (hg if hi[k] < dims[k] else 0)) for k in range(3)]
        innerg = [slice(c0[k] + (hg if lo[k] > 0 else 0), c0[k] + E - (hg if hi[k] < dims[k] else 0)) for k in range(3)]
        a = sal_thr[innerg[0], innerg[1], innerg[2]].cpu().numpy()
        b = s_o[inner[0], inner[1], inner[2]]
        kept = a != 0
        scale = float(np.abs(b).max())
        assert np.all(np.abs(a[kept] - b[kept]) <= 1e-5 * scale), "saliency of kept voxels at %s" % ((z0, y0, x0),)
        # voxels the device zeroed lie below the threshold, up to the tolerance band around it
        assert np.all(b[~kept] < thr + 1e-5 * scale), "thresholded voxels at %s" % ((z0, y0, x0),)


def test_membrane_1024_cubed_crops_equal_oracle(gpu, volume, oracle):
    """BASELINE config 4: `-membrane minima 3 -tv 5 -tv-angle-exponent 4` on 1024^3 (sigma 1.732, top 5 %, sigma_tv 8.66,
    25^3 vote window), every stage fed by the previous device stage as in bench.py."""
    # membranes of the synthetic volume (bench.synth_volume): the tilted plane 0.15 x - 0.1 y + z = 0.35 N and the shell
    # of radius 0.3 N around (0.5, 0.4, 0.5) N
    zp = lambda y, x: int(round(0.35 * N - 0.15 * x + 0.1 * y))
    E = 28
    crops = [
        (0, 0, 0),                                   # corner
        (N - E, N - E, N - E),                       # opposite corner
        (0, 500, 333),                               # z face
        (320 - E // 2, 16 * 20 - E // 2, 16 * 31 - E // 2),   # across a run seam (z = 320) and tile seams in x and y
        (640 - 5, N - E, 7),                         # run seam at z = 640, y face
        (zp(200, 300) - E // 2, 200, 300),           # on the tilted membrane
        (int(0.5 * N + 0.3 * N) - E // 2, int(0.4 * N), int(0.5 * N)),   # on the shell (its top)
    ]
    _check_membrane(gpu, oracle, volume, crops, E)


def test_membrane_1024_cubed_tolerance_mode(gpu, volume, oracle):
    """The same run with tensor voting in TOLERANCE MODE (option tv_fma: fused multiply-adds, mirror-paired sender planes,
    csrc/tv_box.hip): vote tensors within 1e-5 of each crop's scale -- corners, run and tile seams, on the membranes -- and
    the post-vote score within 1e-5 as before."""
    zp = lambda y, x: int(round(0.35 * N - 0.15 * x + 0.1 * y))
    E = 28
    crops = [
        (0, 0, 0),
        (N - E, N - E, N - E),
        (320 - E // 2, 16 * 20 - E // 2, 16 * 31 - E // 2),
        (640 - 5, N - E, 7),
        (zp(200, 300) - E // 2, 200, 300),
        (int(0.5 * N + 0.3 * N) - E // 2, int(0.4 * N), int(0.5 * N)),
    ]
    _check_membrane(gpu, oracle, volume, crops, E, ridge_crops=0, tv_fma=1)


def test_config5_plane_block_crops_equal_oracle(gpu, oracle):
    """BASELINE config 5's per-GPU regime: 2048 x 2048 planes (4 times the tiles of the 1024^3 case per plane, 16 MB
    planes) -- a block of 160 planes of the 2048 x 2048 x 4096 volume, generated from their global z like a slab of the
    8-GPU run (bench.synth_volume with z_offset), through the same stages; crops at corners, faces, across unit seams
    and on the tilted membrane (0.15 x - 0.1 y + z = 0.35 * 4096) against the oracle."""
    import bench
    torch, dev, ctx = gpu
    ctx.trim()     # the 1024^3 tests' workspace is not needed any more
    torch.cuda.empty_cache()
    nzb, n2, z_off, nzg = 160, 2048, 1370, 4096
    src = bench.synth_volume(torch, ctx, (nzb, n2, n2), dev, seed=12345, z_offset=z_off, nz_global=nzg)
    torch.cuda.synchronize()
    E = 28
    zp = lambda y, x: int(round(0.35 * nzg - 0.15 * x + 0.1 * y)) - z_off     # local plane of the tilted membrane
    assert 20 < zp(1000, 600) < nzb - 40
    crops = [
        (0, 0, 0),
        (nzb - E, n2 - E, n2 - E),
        (64 - E // 2, 32 * 40 - E // 2, 8 * 201 - E // 2),        # across run seams (z = 64) and tile seams in y and x
        (96 - 5, n2 - E, 1024 - E // 2),                          # run seam at z = 96, y face, the middle of a row
        (zp(1000, 600) - E // 2, 1000, 600),                      # on the tilted membrane
        (nzb - E, 3, n2 - E),
    ]
    _check_membrane(gpu, oracle, src, crops, E, dense=0.09, ridge_crops=3)


def test_blob_1024_cubed_crops_equal_oracle(gpu, volume, oracle):
    """BASELINE config 3: `-blob-s all out 2.0 4.0 1.066` (12 scales) on 1024^3: the blobs found inside crops at a corner,
    on faces and in the interior equal the oracle's on crop + halo, scores bit for bit."""
    import bench
    from visfd_amd import api, pipeline
    torch, dev, ctx = gpu
    src = volume
    sig = pipeline.cli_blob_sigmas(*bench.BLOB)
    assert len(sig) == 12
    r = api.ratio_from_threshold(0.03)
    mins, maxs = pipeline.blob_detect(ctx, src, sig)
    assert len(mins) > 100000 and len(maxs) > 100000
    halo = int(np.floor(r * float(sig[-1]) * 1.01)) + 2      # widest window + the 3x3x3 neighbourhood
    E = 40
    crops = [(0, 0, 0), (N - E, N - E, N - E), (0, 500, 700), (300, N - E, 0), (512 - 20, 512 - 20, 512 - 20),
             (700, 123, N - E)]
    total = 0
    for (z0, y0, x0) in crops:
        c0 = np.array([z0, y0, x0])
        lo = np.maximum(c0 - halo, 0)
        hi = np.minimum(c0 + E + halo, N)
        sub = _crop(src, lo, hi)
        wmin, wmax = oracle.blob_dog(sub, sig, None, None, 0.02, r, np.inf, -np.inf, False)
        o = c0 - lo
        for got, want, asc in ((mins, wmin, True), (maxs, wmax, False)):
            inside = np.ones(len(got), bool)
            keep = np.ones(len(want), bool)
            for k, col in ((0, 2), (1, 1), (2, 0)):     # k: z, y, x axis; rows are x, y, z, sigma, score
                inside &= (got[:, col] >= c0[k]) & (got[:, col] < c0[k] + E)
                keep &= (want[:, col] >= o[k]) & (want[:, col] < o[k] + E)
            a = got[inside].copy()
            a[:, 0] -= lo[2]; a[:, 1] -= lo[1]; a[:, 2] -= lo[0]
            b = want[keep]
            total += len(b)
            assert_bits_equal(volgen.sort_blobs(a, asc), volgen.sort_blobs(b, asc), "blobs in the crop at %s" % ((z0, y0, x0),))
    assert total > 50


def test_exact_mode_eigen_census(gpu, volume, oracle):
    """Exact mode IS the reference's arithmetic in the ridge stage: with eig_f32 = 0 (the default) the device's float
    eigenvalues / saliencies equal the oracle's bit for bit in (almost) every voxel -- the device's and glibc's double
    libm differ at ~1e-16, which disappears in the float store -- whereas the single-precision angle (eig_f32 = 1, a
    tolerance option) moves most of them by an ulp.  Counted here on tests/golden/eigen.npz and on the 1024^3 bench
    volume: values bit-equal to the oracle with and without the option, and voxels on the other side of the top-5 % cut.
    The census is written to gpurun_out/r4_eig_census.txt (copied to profiles/)."""
    import bench
    from conftest import golden
    from visfd_amd import api
    torch, dev, ctx = gpu
    lines = []
    g = golden("eigen")
    mats = g["mats"]
    for mode in (0, 1):
        with ctx.options(eig_f32=mode):
            for oname, order in (("inc", 0), ("dec", 1)):
                d = ctx.diagonalize(mats, order)
                ref = g["diag_" + oname]
                eq = d[:, :3].view(np.uint32) == ref[:, :3].view(np.uint32)
                lines.append("golden eigen.npz  eig_f32=%d order=%s: %d of %d float eigenvalues bit-equal to the reference (%.4f)" % (
                    mode, oname, int(eq.sum()), eq.size, eq.mean()))
                if mode == 0:
                    assert eq.mean() > 0.995, lines[-1]
    # the bench volume: scores of the whole volume in both modes, the cut each selects, and crops against the oracle
    P = bench.MEMBRANE
    sigma, ratio, order = P["sigma"], api.ratio_from_threshold(0.03), api.DECREASING_EIVALS
    sal = {}
    thr = {}
    for mode in (0, 1):
        s = torch.empty_like(volume)
        sm = torch.empty_like(volume)
        with ctx.options(eig_f32=mode):
            ctx.ridge_scores_dev(volume, s, sm, sigma, ratio, order)
        ctx.synchronize()
        del sm
        raw = s.clone()
        thr[mode] = ctx.threshold_fraction_dev(s, P["best_fraction"])
        ctx.synchronize()
        sal[mode] = (raw, s != 0)
        del s
    nvox = volume.numel()
    same_bits = int((sal[0][0].view(torch.int32) == sal[1][0].view(torch.int32)).sum().item())
    flips = int((sal[0][1] != sal[1][1]).sum().item())
    lines.append("bench volume 1024^3: saliencies bit-equal between eig_f32=0 and =1: %d of %d (%.4f); thresholds %.9g / %.9g; "
                 "voxels on different sides of the top-5 %% cut: %d (of %d kept)" % (
                     same_bits, nvox, same_bits / nvox, thr[0], thr[1], flips, int(sal[0][1].sum().item())))
    hg = int(math.floor(np.float32(sigma) * np.float32(ratio))) + 1
    E = 48
    dims = np.array(volume.shape)
    tot = {0: [0, 0, 0], 1: [0, 0, 0]}
    for (z0, y0, x0) in [(0, 0, 0), (500, 300, 700), (N - E, N - E, N - E), (int(0.35 * N) - 20, 100, 100)]:
        c0 = np.array([z0, y0, x0])
        lo = np.maximum(c0 - hg, 0)
        hi = np.minimum(c0 + E + hg, dims)
        _, hess = oracle.calc_hessian(_crop(volume, lo, hi), np.float32(sigma), ratio)
        s_o, _ = oracle.hessian_saliency(hess, po.ORDER_DECREASING)
        o = c0 - lo
        inner = [slice(o[k] + (hg if lo[k] > 0 else 0), o[k] + E - (hg if hi[k] < dims[k] else 0)) for k in range(3)]
        innerg = [slice(c0[k] + (hg if lo[k] > 0 else 0), c0[k] + E - (hg if hi[k] < dims[k] else 0)) for k in range(3)]
        b = s_o[inner[0], inner[1], inner[2]]
        for mode in (0, 1):
            a = sal[mode][0][innerg[0], innerg[1], innerg[2]].cpu().numpy()
            kept = sal[mode][1][innerg[0], innerg[1], innerg[2]].cpu().numpy()
            tot[mode][0] += int((a.view(np.uint32) == b.view(np.uint32)).sum())
            tot[mode][1] += a.size
            tot[mode][2] += int((kept != (b >= np.float32(thr[0]))).sum())
    for mode in (0, 1):
        lines.append("bench volume, 4 crops of %d^3 against the oracle, eig_f32=%d: %d of %d saliencies bit-equal (%.5f); voxels whose side of the "
                     "cut (threshold of the exact run) differs from the oracle's: %d" % (E, mode, tot[mode][0], tot[mode][1],
                                                                                        tot[mode][0] / tot[mode][1], tot[mode][2]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r4_eig_census.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    assert tot[0][0] / tot[0][1] > 0.995, "exact mode: device saliencies are not the oracle's bits"
    assert tot[0][2] == 0, "exact mode: voxels on the wrong side of the cut"
