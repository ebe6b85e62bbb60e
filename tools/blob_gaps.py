"""Development aid: one blob-detection stage of the bench workload under rocprofv3 --kernel-trace; prints the busy and
idle time between the first and the last kernel of the timed stage (run:  rocprofv3 --kernel-trace --output-format csv
-d gpurun_out/gaps -- python3 tools/blob_gaps.py ; then  python3 tools/blob_gaps.py report gpurun_out/gaps)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "report":
    rows = list(csv.DictReader(open(glob.glob(os.path.join(sys.argv[2], "*", "*kernel_trace.csv"))[0])))
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    rows.sort()
    # the timed stage = everything after the LAST synthetic-volume kernel... simpler: the last 2/3 of the blob kernels:
    # take the launches between the last two "marker" memsets?  Use the blob_candidates launches: the last 10 belong to the timed stage
    cand = [i for i, r in enumerate(rows) if "blob_candidates_kernel" in r[2]]
    last = cand[-10:]
    # stage = from the first Gaussian before the first of these candidates (2 scales earlier: 4 Gaussians) to the last verify after
    i0 = last[0]
    g = 0
    while i0 > 0 and g < 6:      # back over the Gaussians of the first three scales (each 2 launches, wide ones 6)
        i0 -= 1
        if "gauss_fused_kernel" in rows[i0][2]:
            g += 1
    i1 = last[-1] + 1
    seg = rows[i0:i1 + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    busy = 0
    cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("blob stage: %d launches over %.2f ms, busy %.2f ms, idle %.2f ms" % (len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    gaps = sorted(((seg[i + 1][0] - max(r[1] for r in seg[:i + 1])) / 1e3, seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1))
    for gp in gaps[-8:]:
        print("  gap %.1f us  after %s  before %s" % gp)
    sys.exit(0)

import torch  # noqa: E402
from visfd_amd import api, pipeline  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
src = bench.synth_volume(torch, ctx, (1024, 1024, 1024), dev, seed=12345)
sig = pipeline.cli_blob_sigmas(*bench.BLOB)
for _ in range(2):
    mins, maxs = pipeline.blob_detect(ctx, src, sig)
torch.cuda.synchronize()
print(len(mins), len(maxs))
ctx.close()
