"""Quick per-stage timing on one GPU (development aid; bench.py is the contract)."""
import argparse
import sys
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ts.append(e0.elapsed_time(e1))
    return min(ts), wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--stages", default="copy,gauss")
    ap.add_argument("--h", type=str, default="5")
    ap.add_argument("--synth", action="store_true", help="use bench.py's synthetic volume (membranes + blobs) instead of noise")
    a = ap.parse_args()
    n = a.n
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream()  # a real (non-null) stream shared by torch and the library
    torch.cuda.set_stream(stream)
    ctx = api.Context(0, stream.cuda_stream)
    g = torch.Generator(device=dev).manual_seed(1)
    if a.synth:
        import bench
        src = bench.synth_volume(torch, ctx, (n, n, n), dev, seed=12345)
    else:
        src = torch.empty((n, n, n), device=dev, dtype=torch.float32)
        for z in range(0, n, 128):   # in slabs: randn's temporaries stay small at 2048^3
            src[z:z + 128] = torch.randn((min(128, n - z), n, n), device=dev, generator=g) * 100 + 1000
    dst = torch.empty_like(src)
    nvox = n ** 3
    stages = a.stages.split(",")
    if "copy" in stages:
        tmin, tavg = timeit(lambda: dst.copy_(src))
        print("copy   %d^3: %.3f ms (wall %.3f) %.1f GB/s (8 B/vox)" % (n, tmin, tavg, 8 * nvox / tmin / 1e6))
    if "gauss" in stages:
        for h in [int(x) for x in str(a.h).split("+")]:
            sigma = (h / 2.6,) * 3
            tmin, tavg = timeit(lambda: ctx.gauss_dev(src, dst, sigma, (h, h, h)))
            print("gauss  %d^3 h=%d: %.3f ms (wall %.3f) %.1f Gvox/s  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
                n, h, tmin, tavg, nvox / tmin / 1e6, 8 * nvox / tmin / 1e6, 8 * nvox / tmin / 1e6 / 80))
    if "gauss3" in stages:
        h = int(str(a.h).split("+")[0])
        sigma = (h / 2.6, h / 2.6, h / 2.6)
        tmin, tavg = timeit(lambda: ctx.gauss_dev(src, dst, sigma, (h, h, h + 1)))
        print("gauss 3-pass %d^3 h=%d: %.3f ms  %.1f Gvox/s" % (n, h, tmin, nvox / tmin / 1e6))
    if "log" in stages:
        tmin, tavg = timeit(lambda: ctx.log_dev(src, dst, (2, 2, 2), 0.02, 2.6482))
        print("log    %d^3: %.3f ms  %.1f Gvox/s" % (n, tmin, nvox / tmin / 1e6))
    if "ridge" in stages:
        sal = torch.empty_like(src)
        dirs = torch.empty((3, n, n, n), device=dev)
        tmin, tavg = timeit(lambda: ctx.ridge_saliency_dev(src, sal, dirs, 1.732, 2.6482, 1))
        print("ridge  %d^3: %.3f ms  %.1f Gvox/s" % (n, tmin, nvox / tmin / 1e6))
        s2 = sal.clone()
        tmin, tavg = timeit(lambda: (s2.copy_(sal), ctx.threshold_fraction_dev(s2, 0.05)))
        print("select %d^3: %.3f ms  %.1f Gvox/s" % (n, tmin, nvox / tmin / 1e6))
    if "tv" in stages:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sal = torch.empty_like(src)
        dirs = torch.empty((3, n, n, n), device=dev)
        ctx.ridge_saliency_dev(src, sal, dirs, 1.732, 2.6482, 1)
        thr = ctx.threshold_fraction_dev(sal, 0.05)
        print("salient fraction %.4f thr %g" % (float((sal != 0).float().mean()), thr))
        ten = torch.empty((6, n, n, n), device=dev)
        tmin, tavg = timeit(lambda: ctx.tv_dense_stick_dev(sal, dirs, ten, 8.66, 4, 2 ** 0.5), reps=2, warm=1)
        print("tv     %d^3 h=12: %.3f ms  %.4f Gvox/s" % (n, tmin, nvox / tmin / 1e6))
        s2 = sal.clone()
        tmin, tavg = timeit(lambda: ctx.tensor_saliency_dev(ten, s2, 1))
        print("tvsal  %d^3: %.3f ms  %.1f Gvox/s" % (n, tmin, nvox / tmin / 1e6))
    ctx.close()


if __name__ == "__main__":
    main()
