"""One blob scan (three scales -> one candidates + verify pass) for rocprofv3 runs (development aid)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
g = torch.Generator(device=dev).manual_seed(1)
src = torch.randn((n, n, n), device=dev, generator=g) * 100 + 1000
for _ in range(2):
    ctx.blob_dog_dev(src, np.array([2.0, 2.13, 2.27], np.float32), None, None, 0.02, 2.6482, np.inf, -np.inf, False, 1 << 22)
torch.cuda.synchronize()
ctx.close()
print("done", n)
