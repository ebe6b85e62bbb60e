"""Golden vectors for tests/test_fluctuations.py, generated from the REAL reference (oracle/_ref)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyoracle as po  # noqa: E402
from test_fluctuations import CASES, case_inputs, sigmas  # noqa: E402

R = po.load("ref")
out = {}
for i in range(len(CASES)):
    src, mask, radius, ratio, norm = case_inputs(i)
    sg, r = sigmas(radius, ratio)
    out["case%d" % i] = R.local_fluctuations(src, sg, r, mask, norm)
np.savez_compressed(os.path.join(HERE, "fluctuations.npz"), **out)
print("wrote fluctuations.npz", len(out))
