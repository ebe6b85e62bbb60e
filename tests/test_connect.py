"""LabelConnected (connect.hpp:168-1427) through the C ABI (a host-side function) against the
real reference (oracle/_ref) and the committed golden labels (tests/golden/connect.npz, generated from the
reference by tests/golden/make_golden_connect.py).  Inputs are what the CLI hands over after tensor voting:
the post-vote saliency, the vote tensors and their principal eigenvectors."""
import math
import os

import numpy as np
import pytest

import volgen
from visfd_amd import api

GOLD = os.path.join(os.path.dirname(__file__), "golden", "connect.npz")
ANGLE = math.cos(math.pi * 30 / 180.0)    # -connect-angle 30 (settings.cpp:3077-3086)


def tv_outputs(oracle, shape=(28, 34, 40), seed=11, sigma=1.2, tv_ratio=2.5, fraction=0.2):
    """(saliency, tensor [.,6], direction [.,3]) of the membrane pipeline on a seeded volume, from the oracle."""
    from oracle import pyoracle as po
    src = volgen.membrane_volume(shape, seed=seed)
    ratio = oracle.ratio_from_threshold(0.03)
    _, hess = oracle.calc_hessian(src, np.float32(sigma), ratio, None, want_grad=False)
    sal, dirs = oracle.hessian_saliency(hess, po.ORDER_DECREASING)
    oracle.threshold_fraction(sal, fraction)
    ten = oracle.tv_dense_stick(sal, dirs, float(np.float32(tv_ratio) * np.float32(sigma)), 4, 2.0 ** 0.5)
    post = sal.copy()
    oracle.tensor_saliency(ten, po.ORDER_DECREASING, post)
    _, evects = oracle.evects(ten.reshape(-1, 6), po.ORDER_DECREASING)
    direction = np.ascontiguousarray(evects.reshape(shape + (3, 3))[..., 0, :])
    return post, np.ascontiguousarray(ten), direction


def cases(post):
    thr = float(np.sort(post.ravel())[int(0.9 * post.size)])
    return [
        dict(threshold_saliency=thr),                                                      # saliency only
        dict(threshold_saliency=thr, connectivity=3, sort_by_size=False),
        dict(threshold_saliency=thr, use="vt", threshold_vector_saliency=ANGLE, threshold_vector_neighbor=ANGLE,
             threshold_tensor_saliency=ANGLE, threshold_tensor_neighbor=ANGLE, consider_dot_product_sign=False,
             standardize_directions=True),                                                 # the CLI's call
        dict(threshold_saliency=thr, use="vt", threshold_vector_saliency=0.5, threshold_vector_neighbor=0.7,
             threshold_tensor_saliency=-np.inf, threshold_tensor_neighbor=0.8, consider_dot_product_sign=True),
        dict(threshold_saliency=thr * 0.5, use="v", threshold_vector_saliency=0.9, consider_dot_product_sign=False,
             standardize_directions=True, mask=True),
    ]


# strict angle thresholds split the voted surface into seven islands; the must-link groups name seed voxels of some of
# them (the second group starts 0.4 voxels off a voxel centre: locations are rounded, connect.hpp:862-864)
_STRICT = 0.985
_MUST_LINK = [[(17.0, 21.0, 16.0), (22.0, 17.0, 13.0), (12.0, 19.0, 1.0)], [(19.4, 27.4, 13.4), (20.0, 19.0, 1.0), (20.0, 17.0, 2.0)]]


def must_link_cases(post):
    thr = float(np.sort(post.ravel())[int(0.9 * post.size)])
    base = dict(threshold_saliency=thr, use="vt", threshold_vector_saliency=_STRICT, threshold_vector_neighbor=_STRICT,
                threshold_tensor_saliency=0.9, threshold_tensor_neighbor=0.9, consider_dot_product_sign=False,
                standardize_directions=True)
    w = np.random.default_rng(5).uniform(0.5, 2.0, post.shape).astype(np.float32)
    extras = [dict(), dict(must_link=_MUST_LINK), dict(must_link=_MUST_LINK, must_link_directions=[[2, 0, 1], [2, 1, 0]]),
              dict(voxel_weights=w), dict(voxel_weights=w, must_link=_MUST_LINK), dict(must_link=_MUST_LINK, mask=True)]
    return [dict(base, **e) for e in extras]


def run(lib, post, ten, direction, case):
    kw = dict(case)
    use = kw.pop("use", "")
    m = kw.pop("mask", False)
    mask = volgen.block_mask(post.shape, seed=9) if m else None
    d = direction.copy() if "v" in use else None
    t = ten if "t" in use else None
    labels, k, cm, cs, csal = lib.label_connected(post, mask=mask, direction=d, tensor=t, **kw)
    return labels, k, cm, cs, csal, d


def same(a, b, what):
    assert a[1] == b[1], (what, "n_clusters", a[1], b[1])
    assert np.array_equal(a[0], b[0]), (what, "labels differ at", int((a[0] != b[0]).sum()), "voxels")
    for x, y, name in zip(a[2:5], b[2:5], ("seed positions", "sizes", "seed saliencies")):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (what, name)
    if a[5] is not None:
        assert np.array_equal(a[5].view(np.uint32), b[5].view(np.uint32)), (what, "standardized directions")


def test_trace_product_quirk_pinned(ref):
    """The reference's TraceProductSym3 reads its index table out of row bounds; connect.cpp evaluates the
    value the compiled reference actually returns (diagonal entries only)."""
    rng = np.random.default_rng(1)
    for _ in range(200):
        a, b = rng.normal(0, 3, 6).astype(np.float32), rng.normal(0, 3, 6).astype(np.float32)
        want = (a[0] * b[0] + a[0] * b[1] + a[1] * b[2] + a[1] * b[0] + a[1] * b[1] + a[2] * b[2] + a[2] * b[1]
                + a[2] * b[2] + a[0] * b[0])
        assert np.float32(ref.trace_product_sym3(a, b)) == np.float32(want)


def test_label_connected_vs_reference(oracle, ref):
    post, ten, direction = tv_outputs(oracle)
    for i, case in enumerate(cases(post)):
        got, want = run(api, post, ten, direction, case), run(ref, post, ten, direction, case)
        assert want[1] > 0
        same(got, want, "case %d" % i)


def test_must_link_and_voxel_weights_vs_reference(oracle, ref):
    """LabelConnected's remaining optional arguments (connect.hpp:829-1045 and :1154-1290): must-link groups merge the
    islands nearest to the given locations (with and without explicit direction pairs, polarity bookkeeping included),
    voxel weights change the size ranking and the outward orientation -- labels, sizes, seeds and standardized
    directions bit for bit against the compiled reference."""
    post, ten, direction = tv_outputs(oracle)
    counts = []
    for i, case in enumerate(must_link_cases(post)):
        got, want = run(api, post, ten, direction, case), run(ref, post, ten, direction, case)
        same(got, want, "must-link case %d" % i)
        counts.append(got[1])
    assert counts[0] >= 5 and counts[1] < counts[0] and counts[3] == counts[0]     # merges happened; weights merge nothing


def test_label_connected_golden(oracle):
    g = np.load(GOLD)
    post, ten, direction = tv_outputs(oracle)
    import zlib
    assert zlib.crc32(post.tobytes()) == int(g["post_crc"]), "oracle pipeline changed: regenerate goldens"
    for i, case in enumerate(cases(post) + must_link_cases(post)):
        got = run(api, post, ten, direction, case)
        assert got[1] == int(g["n%d" % i])
        assert np.array_equal(got[0], g["labels%d" % i]), "case %d" % i
        if got[5] is not None:
            assert zlib.crc32(got[5].tobytes()) == int(g["dir%d_crc" % i]), "standardized directions, case %d" % i


def test_principal_directions_host_equal_reference_arithmetic(oracle):
    """The host eigen path (eigen3.hpp compiled for the CPU) is bit-identical to the CPU restatement, which is
    bit-identical to the reference (tests/test_oracle_vs_ref.py)."""
    from oracle import pyoracle as po
    _, ten, direction = tv_outputs(oracle)
    for order in (po.ORDER_DECREASING, po.ORDER_INCREASING):
        _, ev = oracle.evects(ten.reshape(-1, 6), order)
        want = np.ascontiguousarray(ev.reshape(ten.shape[:-1] + (3, 3))[..., 0, :])
        got = api.principal_directions_host(ten, order)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        want_sal = np.zeros(ten.shape[:-1], np.float32)
        oracle.tensor_saliency(ten, order, want_sal)
        got_sal = api.tensor_saliency_host(ten, order, np.zeros(ten.shape[:-1], np.float32))
        assert np.array_equal(got_sal.view(np.uint32), want_sal.view(np.uint32))
    rng = np.random.default_rng(4)
    rnd = rng.normal(0, 5, (5000, 6)).astype(np.float32)
    rnd[:50, 3:] = 0                      # diagonal matrices
    rnd[50:60] = 0                        # zero matrices
    rnd[60:70, :3] = 2.0; rnd[60:70, 3:] = 0   # multiples of the identity
    _, ev = oracle.evects(rnd, po.ORDER_DECREASING)
    got = api.principal_directions_host(rnd, po.ORDER_DECREASING)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(ev[:, 0, :]).view(np.uint32))


def test_label_connected_argument_checks():
    s = np.zeros((2, 8, 8), np.float32)
    with pytest.raises(api.VisfdHipError):
        api.label_connected(s, 0.0)


def test_diagonalize_sym3_float_equals_reference(ref):
    """The float instantiation of the 3x3 eigen solver (used by the surface-point export) against the compiled
    reference, bit for bit, for every eigenvalue order, on general, diagonal, degenerate and tiny/huge matrices."""
    rng = np.random.default_rng(8)
    a = rng.normal(0, 3, (3000, 3, 3)).astype(np.float32)
    m = ((a + np.swapaxes(a, 1, 2)) / 2).astype(np.float32)
    m[:40] *= np.float32(1e-12)
    m[40:80] *= np.float32(1e12)
    for k in range(80, 120):
        m[k] = np.diag(rng.normal(0, 2, 3)).astype(np.float32)
    m[120:130] = 0
    for k in range(130, 140):
        m[k] = np.eye(3, dtype=np.float32) * np.float32(rng.normal())
    v = rng.normal(size=(20, 3)).astype(np.float32)
    for k in range(20):
        m[140 + k] = np.outer(v[k], v[k]).astype(np.float32)      # rank one: two equal eigenvalues
    for order in range(4):
        gv, ge = api.diagonalize_sym3_f32_host(m, order)
        wv, we = ref.diagonalize_sym3_f32(m, order)
        assert np.array_equal(gv.view(np.uint32), wv.view(np.uint32)), "eigenvalues, order %d" % order
        assert np.array_equal(ge.view(np.uint32), we.view(np.uint32)), "eigenvectors, order %d" % order


def test_flat_sym3_host_helpers_match_reference(ref):
    """DiagonalizeFlatSym3 / ConvertFlatSym2Evects3 on the host: eigenvalues, Shoemake triple and all
    three eigenvector rows bit for bit against the compiled reference, both orders, including diagonal,
    degenerate and rank-one matrices."""
    from visfd_amd import api
    rng = np.random.default_rng(77)
    m = rng.normal(0.0, 1.0, (300, 6)).astype(np.float32)
    m[200:220, 3:] = 0.0                                     # diagonal matrices
    m[220:240] = 0.0
    m[220:240, :3] = rng.normal(0.0, 1.0, (20, 1)).astype(np.float32)   # multiples of the identity
    v = rng.normal(0.0, 1.0, (20, 3)).astype(np.float32)
    for k in range(20):                                      # rank one: two equal eigenvalues
        o = np.outer(v[k], v[k]).astype(np.float32)
        m[240 + k] = [o[0, 0], o[1, 1], o[2, 2], o[0, 1], o[1, 2], o[0, 2]]
    m[260:280] *= np.float32(1e12)
    m[280:300] *= np.float32(1e-12)
    for order in (0, 1):
        want = ref.diagonalize(m, order)
        got = api.diagonalize_flat_sym3_host(m, order)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "DiagonalizeFlatSym3, order %d" % order
        gv, ge = api.convert_flat_sym2_evects3_host(m, order)
        wv, we = ref.convert_flat_sym2_evects3(m, order)
        assert np.array_equal(gv.view(np.uint32), wv.view(np.uint32)), "eigenvalues, order %d" % order
        assert np.array_equal(ge.view(np.uint32), we.view(np.uint32)), "eigenvector rows, order %d" % order
