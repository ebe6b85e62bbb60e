"""Development aid: the tolerance-mode voting kernel against the exact kernel on random sparse fields, several option sets,
repeated -- how many voxels are NaN / out of tolerance and where."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402


def field(shape, seed, frac=0.05):
    rng = np.random.default_rng(seed)
    sal = rng.random(shape, dtype=np.float32)
    sal[rng.random(shape) > frac] = 0.0
    d = rng.standard_normal(shape + (3,)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    return sal, np.ascontiguousarray(d)


c = api.Context(0)
bad = 0
for shape, sigma in (((20, 22, 26), 3.2), ((20, 37, 45), 8.66), ((41, 33, 70), 3.0), ((9, 20, 33), 1.0), ((12, 40, 50), 11.0)):
    sal, d = field(shape, 7)
    for ex in (4, 2):
        with c.options(tv_fma=0):
            ref = c.tv_dense_stick(sal, d, sigma, ex, 2.0 ** 0.5)
        scale = np.abs(ref).max()
        for opts in ({}, {"tv_max_wg": 3}, {"tv_max_wg": 1, "tv_zrun": 3}, {"tv_no_replay": 1}, {"tv_zrun": 1}, {"tv_zrun": 5}):
            for rep in range(3):
                with c.options(tv_fma=1, **opts):
                    got = c.tv_dense_stick(sal, d, sigma, ex, 2.0 ** 0.5)
                nan = np.isnan(got).any(-1)
                err = np.abs(np.nan_to_num(got) - ref).max(-1) / scale
                off = err > 1e-5
                if nan.any() or off.any():
                    bad += 1
                    zz, yy, xx = np.nonzero(nan | off)
                    print("BAD", shape, sigma, ex, opts, "rep", rep, "nan", int(nan.sum()), "off", int(off.sum()), "max err %.3g" % err.max(),
                          "z", sorted(set(zz.tolist()))[:12], "y", sorted(set(yy.tolist()))[:12], "x", sorted(set(xx.tolist()))[:12])
    print("done", shape, sigma, flush=True)
print("bad cases:", bad)
c.close()
