"""One tensor-voting launch (256^3, 5 % salient, h=12) for rocprofv3 --pmc runs (development aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
g = torch.Generator(device=dev).manual_seed(1)
if len(sys.argv) > 2 and sys.argv[2] == "synth":
    import bench
    src = bench.synth_volume(torch, ctx, (n, n, n), dev, seed=12345)
else:
    src = torch.randn((n, n, n), device=dev, generator=g) * 100 + 1000
sal = torch.empty_like(src)
if n >= 512:   # calibration launches of known traffic for FETCH_SIZE / WRITE_SIZE (see tools/pmc_traffic.py)
    import numpy as np
    torch.cuda.synchronize()
    ctx.apply_threshold_dev(src, -np.inf)
    torch.cuda.synchronize()
    ctx.apply_threshold_dev(sal, np.inf)
    torch.cuda.synchronize()
dirs = torch.empty((3, n, n, n), device=dev)
ctx.ridge_saliency_dev(src, sal, dirs, 1.732, 2.6482, 1)
ctx.threshold_fraction_dev(sal, 0.05)
ten = torch.empty((6, n, n, n), device=dev)
for _ in range(2):
    ctx.tv_dense_stick_dev(sal, dirs, ten, 8.66, 4, 2 ** 0.5)
torch.cuda.synchronize()
ctx.close()
print("done", n)
