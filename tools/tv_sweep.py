"""Development aid: time the tolerance-mode tensor-voting launch of the bench workload under several values of one option.

    python tools/tv_sweep.py tv_zrun 32 48 64 96"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402
import bench  # noqa: E402

n = 1024
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
ctx.set_option("tv_fma", int(os.environ.get("TV_MODE", "1")))
name = sys.argv[1]
for v in sys.argv[2:]:
    ctx.set_option(name, int(v))
    ts = []
    for _ in range(3):
        e0.record()
        ctx.tv_dense_stick_dev(sal, dirs, ten, 8.660254, 4, 2 ** 0.5)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("%s=%s: %s ms  sum|T|=%.9g" % (name, v, " ".join("%.1f" % t for t in ts[1:]), float(ten.abs().sum(dtype=torch.float64))), flush=True)
ctx.close()
