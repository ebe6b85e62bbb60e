"""BinArray3D / UnbinArray3D (resample.hpp:53-166): CPU restatement against the real
reference and the committed golden vectors; HIP kernels against the restatement through the C ABI."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal

GOLD = os.path.join(os.path.dirname(__file__), "golden", "resample.npz")

# (source shape [nz,ny,nx], binned shape, offset (x,y,z) or None)
CASES = [
    ((12, 16, 20), (6, 8, 10), None),           # 2x2x2
    ((13, 16, 21), (6, 8, 10), (1, 0, 1)),
    ((13, 17, 23), (6, 8, 11), None),           # trailing voxels dropped
    ((13, 17, 23), (4, 5, 7), (2, 1, 0)),       # 3x3x3 with an offset
    ((8, 14, 11), (8, 4, 3), (1, 2, 0)),        # anisotropic bins 3,3,1
    ((16, 16, 16), (4, 4, 4), None),            # 4x4x4
    ((13, 13, 13), (2, 2, 2), (1, 1, 1)),       # 6x6x6 (general kernel)
    ((7, 9, 11), (7, 9, 11), None),             # bin 1: copy
]


@pytest.fixture(scope="module")
def ctx():
    from visfd_amd import api
    c = api.Context(0)
    yield c
    c.close()


def vol(shape, seed):
    return (np.random.default_rng(seed).normal(1000, 100, shape)).astype(np.float32)


def test_oracle_resample_golden(oracle):
    g = np.load(GOLD)
    for i, (ss, ds, off) in enumerate(CASES):
        src = vol(ss, 10 + i)
        b = oracle.bin_array3d(src, ds, off)
        assert_bits_equal(b, g["bin%d" % i], "bin case %d" % i)
        assert_bits_equal(oracle.unbin_array3d(b, ss, off), g["unbin%d" % i], "unbin case %d" % i)


def test_oracle_resample_vs_reference(oracle, ref):
    for i, (ss, ds, off) in enumerate(CASES):
        src = vol(ss, 50 + i)
        b = oracle.bin_array3d(src, ds, off)
        assert_bits_equal(b, ref.bin_array3d(src, ds, off), "bin case %d" % i)
        assert_bits_equal(oracle.unbin_array3d(b, ss, off), ref.unbin_array3d(b, ss, off), "unbin case %d" % i)
    for lib in (oracle, ref):
        with pytest.raises(ValueError):
            lib.bin_array3d(vol((8, 8, 8), 1), (4, 4, 4), (2, 0, 0))     # offset == bin size


@pytest.mark.gpu
def test_gpu_resample_parity(ctx, oracle):
    from visfd_amd import api
    for i, (ss, ds, off) in enumerate(CASES):
        src = vol(ss, 90 + i)
        want = oracle.bin_array3d(src, ds, off)
        got = ctx.bin_array3d(src, ds, off)
        assert_bits_equal(got, want, "bin case %d" % i)
        assert_bits_equal(ctx.unbin_array3d(got, ss, off), oracle.unbin_array3d(want, ss, off), "unbin case %d" % i)
    with pytest.raises(api.VisfdHipError):
        ctx.bin_array3d(vol((8, 8, 8), 1), (4, 4, 4), (2, 0, 0))
    with pytest.raises(api.VisfdHipError):
        ctx.bin_array3d(vol((4, 4, 4), 1), (8, 8, 8))                     # "binned" image larger than the source
    with pytest.raises(api.VisfdHipError):
        ctx.bin_array3d(vol((8, 8, 8), 1), (4, 4, 4), (1, 0, 0))         # window would leave the source (reference: UB)


@pytest.mark.gpu
def test_gpu_resample_device_face_large(ctx, oracle):
    """Device face on volumes big enough for many workgroups."""
    import torch
    dev = torch.device("cuda:0")
    small = vol((40, 48, 56), 3)
    big = torch.empty((80, 96, 112), device=dev)
    ctx.unbin_array3d_dev(torch.from_numpy(small).to(dev), big)
    src = vol((81, 97, 113), 4)
    back = torch.empty((40, 48, 56), device=dev)
    ctx.bin_array3d_dev(torch.from_numpy(src).to(dev), back, (1, 0, 1))
    ctx.synchronize()
    assert_bits_equal(big.cpu().numpy(), oracle.unbin_array3d(small, (80, 96, 112)), "unbin dev")
    assert_bits_equal(back.cpu().numpy(), oracle.bin_array3d(src, (40, 48, 56), (1, 0, 1)), "bin dev")
