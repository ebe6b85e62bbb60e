"""Golden vectors for the blob list post-processing tests (tests/test_blob_post.py), generated from
the REAL reference templates through oracle/_ref (run in the build container; only the .npz travels)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyoracle as po  # noqa: E402
from test_blob_post import CASES, random_blobs  # noqa: E402

R = po.load("ref")
out = {}
for i, (seed, n, extent, ties, sep, ml, ms, crit, scale) in enumerate(CASES):
    c, d, s = random_blobs(seed, n, extent, ties)
    oc, od, os_ = R.discard_overlapping_blobs(c, d, s, sep, ml, ms, crit, scale)
    out["ov%d_c" % i], out["ov%d_d" % i], out["ov%d_s" % i] = oc, od, os_
    for crit2 in range(1, 5):
        for asc in (True, False):
            sc, sd, ss, sp = R.sort_blobs(c, d, s, crit2, asc)
            key = "so%d_%d_%d" % (i, crit2, int(asc))
            out[key + "_c"], out[key + "_d"], out[key + "_s"], out[key + "_p"] = sc, sd, ss, sp.astype(np.int64)
rng = np.random.default_rng(77)
mask = (rng.random((24, 20, 28)) > 0.4).astype(np.float32)
mc = np.stack([rng.uniform(0, 27, 200), rng.uniform(0, 19, 200), rng.uniform(0, 23, 200)], 1).astype(np.float32)
mc[::3] = np.floor(mc[::3]) + 0.5   # exercise the floor(x + 0.5) rule at .5
mc = np.clip(mc, 0, [27.4, 19.4, 23.4]).astype(np.float32)
md = rng.uniform(1, 9, 200).astype(np.float32)
ms_ = rng.normal(0, 10, 200).astype(np.float32)
out["mk_in_c"], out["mk_in_d"], out["mk_in_s"], out["mk_mask"] = mc, md, ms_, mask
out["mk_c"], out["mk_d"], out["mk_s"] = R.discard_masked_blobs(mc, md, ms_, mask)
args = []
for _ in range(500):
    ri, rj = rng.uniform(0.5, 20, 2)
    args.append((rng.uniform(0, 1.2 * (ri + rj)), ri, rj))
args += [(0.0, 3.0, 5.0), (2.0, 3.0, 5.0), (8.0, 3.0, 5.0), (5.0, 5.0, 5.0)]
out["sph_args"] = np.array(args, np.float32)
out["sph_out"] = np.array([R.sphere_overlap(*row) for row in out["sph_args"]], np.float32)
np.savez_compressed(os.path.join(HERE, "blob_post.npz"), **out)
print("wrote blob_post.npz:", len(out), "arrays; kept counts:", [len(out["ov%d_d" % i]) for i in range(len(CASES))])
