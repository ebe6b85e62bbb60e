"""The N>1 (Z-slab) path on CPU: world_size 2 and 3 over gloo, oracle arithmetic plugged in for the
kernels (tests/oracle_ops.py).  Slab results must equal the single-volume oracle results bit-for-bit:
halo exchange, ghost handling at true vs interior faces, the distributed exact radix select and the
blob-list merge are what is being tested."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import volgen
from conftest import assert_bits_equal

SHAPE = (30, 18, 20)
SIGMA = 1.2
TV_RATIO = 2.0
FRACTION = 0.15
GHOST = 6
BLOB_SIGMAS = np.array([1.0, 1.25, 1.55, 1.9], np.float32)


def _worker(rank, world, store_path, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_ops import OracleOps
    from visfd_amd import slab
    dist.init_process_group("gloo", init_method="file://" + store_path, rank=rank, world_size=world)
    try:
        ops = OracleOps()
        full = torch.from_numpy(volgen.membrane_volume(SHAPE, seed=55))
        L = slab.SlabLayout(SHAPE[0], rank, world, ghost=GHOST)
        shape = (L.nz_local,) + SHAPE[1:]
        src = torch.full(shape, float("nan"))          # ghosts must come from the exchange, not from here
        L.owned(src).copy_(full[L.z0:L.z1])
        sal = torch.zeros(shape)
        dirs = torch.zeros((3,) + shape)
        ten = torch.zeros((6,) + shape)
        thr = slab.membrane_detect_slab(ops, L, src, sal, dirs, ten, SIGMA, TV_RATIO, 4, FRACTION, 0.03, 2.0 ** 0.5)
        src2 = torch.full(shape, float("nan"))
        L.owned(src2).copy_(full[L.z0:L.z1])
        mins, maxs = slab.blob_detect_slab(ops, L, src2, BLOB_SIGMAS, 0.03, 0.02, -5.0, 5.0, False)
        rmins, rmaxs = slab.blob_detect_slab(ops, L, src2, BLOB_SIGMAS, 0.03, 0.02, 0.5, 0.5, True)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), z0=L.z0, z1=L.z1, thr=np.float32(thr),
                 sal=L.owned(sal).numpy(), ten=L.owned(ten).numpy(), mins=mins, maxs=maxs, rmins=rmins, rmaxs=rmaxs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_pipeline_equals_single_volume(oracle, world):
    from oracle import pyoracle as po
    with tempfile.TemporaryDirectory() as tmp:
        store = os.path.join(tmp, "store")
        mp.spawn(_worker, args=(world, store, tmp), nprocs=world, join=True)
        parts = [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(world)]
    # single-volume oracle
    full = volgen.membrane_volume(SHAPE, seed=55)
    ratio = oracle.ratio_from_threshold(0.03)
    _, hess = oracle.calc_hessian(full, SIGMA, ratio, None, want_grad=False)
    sal, dirs = oracle.hessian_saliency(hess, po.ORDER_DECREASING)
    thr = oracle.threshold_fraction(sal, FRACTION)
    sigma_tv = float(np.float32(TV_RATIO) * np.float32(SIGMA))
    ten = oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5)
    oracle.tensor_saliency(ten, po.ORDER_DECREASING, sal)
    assert np.abs(ten).max() > 0
    for p in parts:
        assert np.float32(p["thr"]) == np.float32(thr)
        z0, z1 = int(p["z0"]), int(p["z1"])
        assert_bits_equal(p["sal"], sal[z0:z1], "post-voting saliency of planes %d..%d" % (z0, z1))
        assert_bits_equal(p["ten"], np.ascontiguousarray(np.moveaxis(ten, -1, 0))[:, z0:z1], "vote tensor")
    a, b = oracle.blob_dog(full, BLOB_SIGMAS, None, None, 0.02, ratio, -5.0, 5.0, False)
    ra, rb = oracle.blob_dog(full, BLOB_SIGMAS, None, None, 0.02, ratio, 0.5, 0.5, True)
    assert len(a) + len(b) > 4
    for p in parts:  # every rank holds the merged lists
        assert_bits_equal(volgen.sort_blobs(p["mins"], True), volgen.sort_blobs(a, True), "minima")
        assert_bits_equal(volgen.sort_blobs(p["maxs"], False), volgen.sort_blobs(b, False), "maxima")
        assert_bits_equal(volgen.sort_blobs(p["rmins"], True), volgen.sort_blobs(ra, True), "ratio minima")
        assert_bits_equal(volgen.sort_blobs(p["rmaxs"], False), volgen.sort_blobs(rb, False), "ratio maxima")


def test_layout_covers_volume():
    from visfd_amd import slab
    for nz, world, ghost in ((30, 2, 5), (31, 3, 4), (1024, 8, 12), (17, 1, 3)):
        seen = np.zeros(nz, int)
        for r in range(world):
            L = slab.SlabLayout(nz, r, world, ghost)
            seen[L.z0:L.z1] += 1
            assert L.lo == max(0, L.z0 - ghost) and L.hi == min(nz, L.z1 + ghost)
            assert L.own1 - L.own0 == L.z1 - L.z0
        assert (seen == 1).all()
