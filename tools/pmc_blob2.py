"""A few blob-detection scales at 1024^3 for rocprofv3 --pmc runs (development aid)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api, pipeline  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
g = torch.Generator(device=dev).manual_seed(1)
src = torch.empty((n, n, n), device=dev)
for z in range(0, n, 128):
    src[z:z + 128] = torch.randn((min(128, n - z), n, n), device=dev, generator=g) * 100 + 1000
sig = pipeline.cli_blob_sigmas(2.0, 4.0, 1.066)[:4]
torch.cuda.synchronize()
mins, maxs = ctx.blob_dog_dev(src, sig, None, None, 0.02, api.ratio_from_threshold(0.03), np.inf, -np.inf, False, cap=1 << 22)
print("done", len(mins), len(maxs))
