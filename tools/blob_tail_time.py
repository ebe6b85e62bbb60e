"""Development aid: where the host time at the end of the blob stage goes -- the C call (visfd_hip_blob_dog_dev: filters, scans,
per-scale list fetch + sort, final merge) against the Python conversion of its result, several repetitions in one process."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api, pipeline  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
src = bench.synth_volume(torch, ctx, (n, n, n), dev, seed=12345)
sig = np.asarray(pipeline.cli_blob_sigmas(*bench.BLOB), np.float32)
orig = api._blobs_to_rows
conv = [0.0]


def timed_conv(arr, k):
    t = time.perf_counter()
    r = orig(arr, k)
    conv[0] += time.perf_counter() - t
    return r


api._blobs_to_rows = timed_conv
if os.environ.get("NOGC"):
    import gc
    gc.disable()
for i in range(reps):
    conv[0] = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mins, maxs = pipeline.blob_detect(ctx, src, sig)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("rep %d: blob_detect %.1f ms, of which Python conversion %.1f ms; %d + %d blobs" % (i, (t1 - t0) * 1e3, conv[0] * 1e3, len(mins), len(maxs)), flush=True)
ctx.close()
