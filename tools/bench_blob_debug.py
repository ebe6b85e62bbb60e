"""Development aid: bench.py's own steps with the host side of the blob stage timed (the C call, the conversion of its result):
    python tools/bench_blob_debug.py [bench.py arguments]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402

acc = {"call": [], "conv": []}
orig_call = api.Context._blob_call
orig_conv = api._blobs_to_rows


def conv(arr, n):
    t = time.perf_counter()
    r = orig_conv(arr, n)
    acc["conv"][-1] += (time.perf_counter() - t) * 1e3
    return r


def call(self, *a, **k):
    acc["conv"].append(0.0)
    t = time.perf_counter()
    r = orig_call(self, *a, **k)
    acc["call"].append((time.perf_counter() - t) * 1e3)
    return r


api.Context._blob_call = call
api._blobs_to_rows = conv
import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
try:
    bench.main()
finally:
    print("blob calls (ms, C + conversion): " + " ".join("%.1f" % c for c in acc["call"]), file=sys.stderr)
    print("of which conversion:             " + " ".join("%.1f" % c for c in acc["conv"]), file=sys.stderr)
