"""The C++ drop-in face on the GPU: tests/shim_gpu_check.cpp is written against the reference's template API
(float*** arrays, CompactMultiChannelImage3D, std::vector result lists), compiled with g++ against
include/visfd_hip.hpp + libvisfd_hip.so and run; every result is compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import volgen
from conftest import ROOT, assert_bits_equal
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

SHAPE = (18, 22, 26)     # nz, ny, nx
TV_SIGMA = 2.3


def _read_records(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            tag = f.read(32)
            if len(tag) < 32:
                break
            n, = struct.unpack("<q", f.read(8))
            out[tag.split(b"\0")[0].decode()] = np.frombuffer(f.read(4 * n), np.float32).copy()
    return out


@pytest.fixture(scope="module")
def shim_run(tmp_path_factory, oracle):
    d = tmp_path_factory.mktemp("shim")
    exe = str(d / "shim_gpu_check")
    libdir = os.path.join(ROOT, "visfd_amd")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "shim_gpu_check.cpp"), "-o", exe, "-L" + libdir, "-lvisfd_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    nz, ny, nx = SHAPE
    src = volgen.blob_volume(SHAPE, seed=901, nblobs=14)
    mask = volgen.block_mask(SHAPE, seed=902)
    rng = np.random.default_rng(903)
    sal = (rng.random(SHAPE) < 0.2).astype(np.float32) * rng.uniform(0.5, 3.0, SHAPE).astype(np.float32)
    dirs = rng.standard_normal(SHAPE + (3,)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True).astype(np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    with open(d / "in.bin", "wb") as f:
        f.write(struct.pack("<iiif", nx, ny, nz, TV_SIGMA))
        for a in (src, mask, sal, dirs):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "shim gpu check ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    return dict(out=_read_records(d / "out.bin"), src=src, mask=mask, sal=sal, dirs=dirs)


def test_apply_separable_and_dog(shim_run, oracle):
    R, src, mask = shim_run["out"], shim_run["src"], shim_run["mask"]
    want, A = oracle.gauss_hw(src, (1.2, 1.5, 0.9), (3, 4, 2), mask, True)
    assert_bits_equal(R["separable"].reshape(SHAPE), want, "ApplySeparable (masked, normalised, three filters)")
    assert np.float32(R["separable_A"][0]) == np.float32(A)
    want, _, _ = oracle.dog(src, (1.0, 1.1, 1.2), (1.6, 1.7, 1.8), (4, 4, 5))
    assert_bits_equal(R["dog"].reshape(SHAPE), want, "ApplyDog")


def test_blob_dog_and_non_max_suppression(shim_run, oracle):
    from visfd_amd import api
    R, src, mask = shim_run["out"], shim_run["src"], shim_run["mask"]
    sig = np.array([1.0, 1.3, 1.7, 2.2], np.float32)
    wmin, wmax = oracle.blob_dog(src, sig, mask, None, 0.02, 2.5, 0.5, 0.5, True)
    assert len(wmin) + len(wmax) > 0
    assert_bits_equal(volgen.sort_blobs(R["blob_min"].reshape(-1, 5), True), volgen.sort_blobs(wmin, True), "BlobDog minima")
    assert_bits_equal(volgen.sort_blobs(R["blob_max"].reshape(-1, 5), False), volgen.sort_blobs(wmax, False), "BlobDog maxima")
    # BlobDogNM = BlobDogD + DiscardOverlappingBlobs (feature_variants.hpp:448-502)
    diam = np.array([3.5, 4.5, 5.9, 7.6], np.float32)
    bmin, bmax = oracle.blob_dog(src, oracle.diameters_to_sigmas(diam), None, None, 0.02, 2.5, 0.9, 0.9, True)
    for got, rows, crit, asc in ((R["nm_min"], bmin, api.SORT_INCREASING, True), (R["nm_max"], bmax, api.SORT_DECREASING, False)):
        rows = volgen.sort_blobs(rows, asc)     # the library returns lists sorted by (scale, z, y, x); scan order too
        d = oracle.sigmas_to_diameters(rows[:, 3].copy())
        c, dd, s = api.discard_overlapping_blobs(rows[:, :3], d, rows[:, 4], 1.0, 1.0, 1.0, crit)
        want = np.concatenate([c, dd[:, None], s[:, None]], axis=1).astype(np.float32)
        g = got.reshape(-1, 5)
        assert len(g) == len(want) and len(want) > 0
        assert_bits_equal(g[np.lexsort(g.T[::-1])], want[np.lexsort(want.T[::-1])], "BlobDogNM list")


def test_calc_hessian_into_compact_container(shim_run, oracle):
    R, src, mask = shim_run["out"], shim_run["src"], shim_run["mask"]
    grad, hess = oracle.calc_hessian(src, np.float32(1.4), 2.5, mask)
    got_h = R["hessian"].reshape(SHAPE + (6,))
    got_g = R["gradient"].reshape(SHAPE + (3,))
    keep = mask != 0
    assert keep.any() and (~keep).any()
    assert_bits_equal(got_h[keep], hess[keep], "Hessian of unmasked voxels")
    assert_bits_equal(got_g[keep], grad[keep], "gradient of unmasked voxels")
    assert np.all(got_h[~keep] == -7.0) and np.all(got_g[~keep] == -7.0)     # no storage / untouched where mask == 0
    assert_bits_equal(R["hessian_copy"], R["hessian"], "CompactMultiChannelImage3D copy constructor")


@pytest.mark.parametrize("name,ms,md,norm,diag", [
    ("tv_plain", False, False, False, False), ("tv_masked", True, True, False, False),
    ("tv_default_args", True, True, True, False), ("tv_norm_dst_only", False, True, True, False),
    ("tv_norm_no_dst", True, False, True, False), ("tv_diag", True, True, False, True)])
def test_tv_dense_stick_with_the_reference_defaults(shim_run, oracle, name, ms, md, norm, diag):
    """TV3D::TVDenseStick through the shim, including normalize = true (the reference's default argument) and
    diagonalize_dest, which run the reference's own arithmetic -- oddities included (feature.hpp:1784-1901)."""
    R, mask, sal, dirs = shim_run["out"], shim_run["mask"], shim_run["sal"], shim_run["dirs"]
    want = oracle.tv_dense_stick(sal, dirs, TV_SIGMA, 4, float(np.sqrt(np.float32(2.0))),
                                 mask if ms else None, mask if md else None, False, norm, diag)
    got = R[name].reshape(SHAPE + (6,))
    keep = (mask != 0) if md else np.ones(SHAPE, bool)
    assert np.abs(want[keep]).max() > 0
    # (the diagonalisation runs on the host in both: the same closed-form solver and the same libm, bit for bit)
    assert_bits_equal(got[keep], want[keep], name)
    if md:
        assert np.all(got[~keep] == -7.0)
