"""Development aid: time the blob-detection stage of the bench workload (12 scales, 1024^3) a few times."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api, pipeline  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
src = bench.synth_volume(torch, ctx, (1024, 1024, 1024), dev, seed=12345)
sig = pipeline.cli_blob_sigmas(*bench.BLOB)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(4):
    e0.record()
    mins, maxs = pipeline.blob_detect(ctx, src, sig)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("%s: blob stage %s ms  (%d minima, %d maxima)" % (os.environ.get("VISFD_HIP_LIB", "default"), " ".join("%.1f" % t for t in ts[1:]),
                                                       len(mins), len(maxs)))
ctx.close()
