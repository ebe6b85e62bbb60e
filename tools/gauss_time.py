"""Development aid: time the unmasked normalised Gaussian at n^3 for a list of half-widths, with and without the option
gauss_3pass (three single-axis kernels), with HIP events on the context's stream.

    python tools/gauss_time.py [n] [h,h,...] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hs = [int(h) for h in (sys.argv[2] if len(sys.argv) > 2 else "8,9,10").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
src = torch.randn((n, n, n), device=dev)
dst = torch.empty_like(src)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for h in hs:
    for three in (0, 1):
        ctx.set_option("gauss_3pass", three)
        ts = []
        for _ in range(reps + 1):
            e0.record()
            ctx.gauss_dev(src, dst, (h / 2.6,) * 3, (h, h, h))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("h=%d gauss_3pass=%d: %s ms  checksum %.9g" % (h, three, " ".join("%.3f" % t for t in ts[1:]), float(dst.double().sum())))
ctx.set_option("gauss_3pass", 0)
ctx.close()
