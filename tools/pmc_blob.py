"""Development aid: the blob stage of the bench workload (BlobDog, a few scales on the 1024^3 synthetic volume) alone, for
rocprofv3 --pmc runs over its kernels (gauss_fused_kernel<6..8>, conv_*_kernel, blob_candidates_kernel, blob_verify_kernel):
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES ... -- python3 tools/pmc_blob.py [n] [first_scale] [scales]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api, pipeline  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
first = int(sys.argv[2]) if len(sys.argv) > 2 else 4
count = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
src = bench.synth_volume(torch, ctx, (n, n, n), dev, seed=12345)
sigmas = np.asarray(pipeline.cli_blob_sigmas(*bench.BLOB), np.float32)[first:first + count]
mins, maxs = pipeline.blob_detect(ctx, src, sigmas)
torch.cuda.synchronize()
print("done", n, sigmas, len(mins), len(maxs))
ctx.close()
