"""Development aid: time the eigen-bound kernels of the membrane stage (ridge scores, directions, post-vote score) at n^3."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
src = bench.synth_volume(torch, ctx, (n, n, n), dev, seed=12345)
sal, sm = torch.empty_like(src), torch.empty_like(src)
dirs = torch.zeros((3, n, n, n), device=dev)
ten = torch.randn((6, n, n, n), device=dev)
r = api.ratio_from_threshold(0.03)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


a = t(lambda: ctx.ridge_scores_dev(src, sal, sm, 1.7320508, r, 1))
thr = ctx.threshold_fraction_dev(sal, 0.05)
b = t(lambda: ctx.ridge_directions_dev(sm, sal, dirs, 1.7320508, 1))
s2 = sal.clone()
c = t(lambda: ctx.tensor_saliency_dev(ten, s2, 1))
print("%s: ridge_scores (incl. gauss) %.2f ms  directions %.2f ms  tensor_saliency %.2f ms  thr %.9g  sum(sal) %.9g" % (
    os.environ.get("VISFD_HIP_LIB", "default"), a, b, c, thr, float(sal.sum(dtype=torch.float64))))
ctx.close()
