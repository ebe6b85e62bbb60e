"""Development aid: the membrane stage of the bench workload (1024^3), three times, for a rocprofv3 --kernel-trace run:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mg -- python3 tools/membrane_gaps.py
    python3 tools/membrane_gaps.py report gpurun_out/mg      # kernels and idle gaps of the last repetition
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "report":
    rows = list(csv.DictReader(open(glob.glob(sys.argv[2] + "/*/*kernel_trace.csv")[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last_tv = max(i for i, r in enumerate(rows) if "tv_box_kernel" in r["Kernel_Name"])
    first = max(i for i, r in enumerate(rows[:last_tv]) if "gauss_fused_kernel<4" in r["Kernel_Name"])
    end = last_tv + 1
    while end < len(rows) and "tensor_saliency" not in rows[end]["Kernel_Name"]:
        end += 1
    seg = rows[first:end + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.3f ms  +gap %7.3f  dur %9.3f  %s" % ((s - t0) / 1e6, (s - prev_end) / 1e6, (e - s) / 1e6, r["Kernel_Name"][:80]))
        busy += e - s
        prev_end = e
    print("stage %.3f ms, kernels %.3f ms, idle %.3f ms" % ((prev_end - t0) / 1e6, busy / 1e6, (prev_end - t0 - busy) / 1e6))
    sys.exit(0)

import torch  # noqa: E402
from visfd_amd import api, pipeline  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
n = 1024
src = bench.synth_volume(torch, ctx, (n, n, n), dev, seed=12345)
sal, scratch = torch.empty_like(src), torch.empty_like(src)
dirs = torch.empty((3, n, n, n), device=dev)
ten = torch.empty((6, n, n, n), device=dev)
with ctx.options(tv_fma=1, gauss_fma=1):
    for _ in range(3):
        pipeline.membrane_detect(ctx, src, sal, dirs, ten, scratch=scratch, **bench.MEMBRANE)
    torch.cuda.synchronize()
ctx.close()
