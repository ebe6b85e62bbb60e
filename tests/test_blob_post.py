"""Blob list post-processing: SortBlobs, DiscardMaskedBlobs, DiscardOverlappingBlobs,
CalcSphereOverlap through the C ABI (host-side functions) against the real reference (oracle/_ref, when
present) and against committed golden vectors (tests/golden/blob_post.npz, generated from the reference by
tests/golden/make_golden_blob_post.py)."""
import os

import numpy as np
import pytest

from visfd_amd import api

GOLD = os.path.join(os.path.dirname(__file__), "golden", "blob_post.npz")


def random_blobs(seed, n, extent=120.0, ties=False):
    rng = np.random.default_rng(seed)
    crds = np.floor(rng.uniform(0, extent, (n, 3))).astype(np.float32)      # detector output: integer voxels
    diam = rng.uniform(3.0, 22.0, n).astype(np.float32)
    score = (rng.normal(0, 50, n)).astype(np.float32)
    if ties:
        score = np.round(score / 25).astype(np.float32) * 25                 # many equal scores
    return crds, diam, score


CASES = [  # (seed, n, extent, ties, min_sep, max_large, max_small, criteria, scale)
    (1, 400, 120.0, False, 1.0, np.inf, np.inf, api.SORT_DECREASING_MAGNITUDE, 6),
    (2, 400, 120.0, True, 0.8, np.inf, np.inf, api.SORT_DECREASING_MAGNITUDE, 6),
    (3, 300, 90.0, False, 0.0, 0.3, np.inf, api.SORT_INCREASING, 6),
    (4, 300, 90.0, True, 0.0, np.inf, 0.5, api.SORT_DECREASING, 4),
    (5, 500, 60.0, False, 1.5, 0.2, 0.6, api.SORT_INCREASING_MAGNITUDE, 6),
    (6, 40, 20.0, False, 1.0, np.inf, np.inf, api.SORT_DECREASING_MAGNITUDE, 6),   # tiny extent: grid of 1-3 cells
    (7, 5, 4.0, False, 1.0, np.inf, np.inf, api.SORT_DECREASING_MAGNITUDE, 6),     # grid with zero cells: nothing collides
    (8, 1, 50.0, False, 1.0, np.inf, np.inf, api.SORT_DECREASING_MAGNITUDE, 6),
]


def _eq(a, b, what):
    assert len(a) == len(b), (what, [len(x) for x in a], [len(x) for x in b])
    for x, y, name in zip(a, b, ("crds", "diameters", "scores", "permutation")):
        x, y = np.asarray(x), np.asarray(y)
        assert x.shape == y.shape, (what, name, x.shape, y.shape)
        if x.dtype == np.float32:
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (what, name)
        else:
            assert np.array_equal(x, y), (what, name)


def test_blob_post_golden():
    g = np.load(GOLD)
    for i, (seed, n, extent, ties, sep, ml, ms, crit, scale) in enumerate(CASES):
        c, d, s = random_blobs(seed, n, extent, ties)
        got = api.discard_overlapping_blobs(c, d, s, sep, ml, ms, crit, scale)
        _eq(got, (g["ov%d_c" % i], g["ov%d_d" % i], g["ov%d_s" % i]), "overlap case %d" % i)
        for crit2 in range(1, 5):
            for asc in (True, False):
                got = api.sort_blobs(c, d, s, crit2, asc)
                key = "so%d_%d_%d" % (i, crit2, int(asc))
                _eq(got[:3] + (got[3].astype(np.int64),), (g[key + "_c"], g[key + "_d"], g[key + "_s"], g[key + "_p"]), key)
    _eq(api.discard_masked_blobs(g["mk_in_c"], g["mk_in_d"], g["mk_in_s"], g["mk_mask"]),
        (g["mk_c"], g["mk_d"], g["mk_s"]), "masked")
    r = g["sph_args"]
    got = np.array([api.sphere_overlap(*row) for row in r], np.float32)
    assert np.array_equal(got.view(np.uint32), g["sph_out"].view(np.uint32))


def test_blob_post_vs_reference(ref):
    for i, (seed, n, extent, ties, sep, ml, ms, crit, scale) in enumerate(CASES):
        c, d, s = random_blobs(seed + 100, n, extent, ties)
        _eq(api.discard_overlapping_blobs(c, d, s, sep, ml, ms, crit, scale),
            ref.discard_overlapping_blobs(c, d, s, sep, ml, ms, crit, scale), "overlap case %d" % i)
        for crit2 in range(0, 5):
            for asc in (True, False):
                a = api.sort_blobs(c, d, s, crit2, asc)
                b = ref.sort_blobs(c, d, s, crit2, asc)
                if crit2 == api.DO_NOT_SORT:
                    _eq(a[:3], b[:3], "no sort")
                else:
                    _eq(a, b, "sort %d %d" % (crit2, asc))
    rng = np.random.default_rng(5)
    for _ in range(2000):
        ri, rj = rng.uniform(0.5, 20, 2)
        rij = rng.uniform(0, ri + rj)
        assert np.float32(api.sphere_overlap(rij, ri, rj)) == np.float32(ref.sphere_overlap(rij, ri, rj))


def test_discard_masked_rejects_out_of_image():
    mask = np.ones((4, 4, 4), np.float32)
    with pytest.raises(api.VisfdHipError):
        api.discard_masked_blobs(np.array([[9, 1, 1]], np.float32), np.ones(1, np.float32), np.ones(1, np.float32), mask)
    c, d, s = api.discard_masked_blobs(np.zeros((0, 3), np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32), mask)
    assert len(d) == 0


def test_reference_blob_scenario():
    """tests/test_blob_detection.sh of the reference: the 11 detected minima, after `-discard-blobs
    -blob-separation 1.1 -minima-threshold -90`, leave 2 blobs."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "blob_rec.npz"))
    mins = g["minima"]              # rows: x, y, z (voxels), sigma, score
    c, d, s = mins[:, :3], g["minima_diam_vox"], mins[:, 4]
    keep = s <= np.float32(-90.0)   # handlers.cpp:516-540 with score_upper_bound = -90
    c, d, s = api.discard_overlapping_blobs(c[keep], d[keep], s[keep], 1.1)
    assert len(d) == 2
    assert s[0] == mins[:, 4].min()
