"""Golden vectors for tests/test_resample.py, generated from the REAL reference (oracle/_ref)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyoracle as po  # noqa: E402
from test_resample import CASES, vol  # noqa: E402

R = po.load("ref")
out = {}
for i, (ss, ds, off) in enumerate(CASES):
    src = vol(ss, 10 + i)
    out["bin%d" % i] = R.bin_array3d(src, ds, off)
    out["unbin%d" % i] = R.unbin_array3d(out["bin%d" % i], ss, off)
np.savez_compressed(os.path.join(HERE, "resample.npz"), **out)
print("wrote resample.npz", len(out))
