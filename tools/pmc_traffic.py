"""HBM traffic of the separable-Gaussian kernel from PMC counters (run under rocprofv3 --pmc FETCH_SIZE
and, in a separate pass, --pmc WRITE_SIZE).  Two calibration launches with known byte counts in the
same access style (dword loads / dword stores, x-contiguous) come first:
  apply_threshold(thr=-inf): reads 4 B/voxel, writes nothing
  apply_threshold(thr=+inf): reads 4 B/voxel, writes 4 B/voxel
then the Gaussian (sigma=2, h=5) itself."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api, pipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
g = torch.Generator(device=dev).manual_seed(1)
src = torch.randn((n, n, n), device=dev, generator=g) * 100 + 1000
dst = torch.empty_like(src)
torch.cuda.synchronize()
ctx.apply_threshold_dev(src, -np.inf)   # calibration A: pure dword reads
torch.cuda.synchronize()
ctx.apply_threshold_dev(dst, np.inf)    # calibration B: dword reads + dword writes (dst becomes zeros)
torch.cuda.synchronize()
for _ in range(3):
    pipeline.gauss(ctx, src, dst, 2.0)
torch.cuda.synchronize()
ctx.close()
print("done", n)
