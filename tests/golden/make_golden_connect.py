"""Golden labels for tests/test_connect.py, generated from the REAL reference (oracle/_ref)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pyoracle as po  # noqa: E402
import test_connect as tc  # noqa: E402

O, R = po.load("oracle"), po.load("ref")
post, ten, direction = tc.tv_outputs(O)
import zlib  # noqa: E402

out = {"post_crc": np.int64(zlib.crc32(post.tobytes()))}
for i, case in enumerate(tc.cases(post) + tc.must_link_cases(post)):
    labels, k, cm, cs, csal, d = tc.run(R, post, ten, direction, case)
    out["labels%d" % i] = labels.astype(np.int32)
    out["n%d" % i] = np.int64(k)
    if d is not None:
        out["dir%d_crc" % i] = np.int64(zlib.crc32(d.tobytes()))
np.savez_compressed(os.path.join(HERE, "connect.npz"), **out)
print("wrote connect.npz; clusters:", [int(out["n%d" % i]) for i in range(len(tc.cases(post)) + len(tc.must_link_cases(post)))])
