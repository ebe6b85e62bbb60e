"""Development aid: the sequence of tests/test_tolerance_modes.py::test_tv_fma_seeded_goldens, repeated; where are the bad voxels?"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import volgen
from conftest import golden
from visfd_amd import api
g = golden("membrane_seeded")
c = api.Context(0)
c.set_option("tv_poison", int(os.environ.get("POISON", "1")))
nbad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    for tag in ("nomask", "mask"):
        m = volgen.block_mask(volgen.MEM_SHAPE, seed=302) if tag == "mask" else None
        sal, dirs = g[tag + "_salthr"], g[tag + "_dir"]
        for opts in ({}, {"tv_max_wg": 3}, {"tv_max_wg": 1, "tv_zrun": 3}, {"tv_no_replay": 1}):
            with c.options(tv_fma=int(os.environ.get("TVM", "1")), **opts):
                for ex in (4, 2):
                    want = g["%s_tensor_e%d" % (tag, ex)]
                    ten = c.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, ex, 2.0 ** 0.5, m, m)
                    nan = np.isnan(ten).any(-1)
                    err = np.abs(np.nan_to_num(ten) - want).max(-1) / np.abs(want).max()
                    if nan.any() or (err > 1e-5).any():
                        nbad += 1
                        zz, yy, xx = np.nonzero(nan | (err > 1e-5))
                        print("BAD rep", rep, tag, opts, ex, "nan", int(nan.sum()), "off", int((err > 1e-5).sum()), "max err %.3g" % err.max(),
                              "zyx", list(zip(zz.tolist(), yy.tolist(), xx.tolist()))[:6], "values", ten[zz[:3], yy[:3], xx[:3]].tolist(), "want", want[zz[:3], yy[:3], xx[:3]].tolist(), flush=True)
                        if nbad > 6: raise SystemExit(0)
                ten = c.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, 4, 2.0 ** 0.5, m, m, curves=True)
print("bad:", nbad)
