"""Development aid: time the tensor-voting launch of the bench workload (1024^3 synthetic volume, 5 % salient, h = 12) in
exact and tolerance mode with the library VISFD_HIP_LIB points at, and print a digest of the result (to compare variants).

    VISFD_HIP_LIB=visfd_amd/_variants/x.so python tools/tv_time.py [n] [reps]"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
src = bench.synth_volume(torch, ctx, (n, n, n), dev, seed=12345)
sal = torch.empty_like(src)
dirs = torch.empty((3, n, n, n), device=dev)
ctx.ridge_saliency_dev(src, sal, dirs, 1.7320508, api.ratio_from_threshold(0.03), 1)
ctx.threshold_fraction_dev(sal, 0.05)
del src
ten = torch.empty((6, n, n, n), device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mode in [int(m) for m in os.environ.get('TV_MODES', '0,1').split(',')]:
    ctx.set_option("tv_fma", mode)
    ts = []
    for _ in range(reps + 1):
        e0.record()
        ctx.tv_dense_stick_dev(sal, dirs, ten, 8.660254, 4, 2 ** 0.5)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    crop = ten[:, n // 3:n // 3 + 40, 100:140, 200:240].contiguous().cpu().numpy()
    print("%s tv_fma=%d: %s ms  sum|T|=%.9g  crop md5 %s" % (os.environ.get("VISFD_HIP_LIB", "default"), mode,
          " ".join("%.1f" % t for t in ts[1:]), float(ten.abs().sum(dtype=torch.float64)), hashlib.md5(crop.tobytes()).hexdigest()[:12]))
ctx.close()
