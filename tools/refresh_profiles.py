"""Copy a rocprofv3 --kernel-trace --stats run of bench.py (gpurun_out/<dir>) into profiles/ (development aid).

    python tools/refresh_profiles.py prof_r01_f bench_r01_f.json bench_r01_f_prof.json
"""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof, bench_json, bench_prof_json = sys.argv[1:4]
src = glob.glob(os.path.join(ROOT, "gpurun_out", prof, "*", "*kernel_stats.csv"))[0]
shutil.copy(src, os.path.join(ROOT, "profiles", "r01_bench_1024_kernel_stats.csv"))
rows = list(csv.DictReader(open(src)))
with open(os.path.join(ROOT, "profiles", "r01_bench_1024_summary.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu"
            "   (MI355X, 1024^3, round 1, final kernels)\n")
    f.write("# 3 pipeline steps + 11 extra Gaussian launches (roofline timing) + synthetic-input generation\n")
    f.write("# gauss_fused_kernel<5,...>: the 11 roofline launches + 3 pipeline launches are plain Gaussians (the roofline\n"
            "#   object's ms_per_launch); the other 12 are second Gaussians of LoG scales, which also read the minuend\n"
            "#   (12 B/voxel), so the average over all calls sits a few percent above ms_per_launch\n")
    f.write("%-100s %6s %12s %10s %7s\n" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in rows[:22]:
        f.write("%-100s %6d %12.3f %10.4f %7s\n" % (r["Name"][:100], int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e6, r["Percentage"]))
shutil.copy(os.path.join(ROOT, "gpurun_out", bench_json), os.path.join(ROOT, "profiles", "r01_bench_1024.json"))
shutil.copy(os.path.join(ROOT, "gpurun_out", bench_prof_json), os.path.join(ROOT, "profiles", "r01_bench_1024_under_rocprof.json"))
print(open(os.path.join(ROOT, "profiles", "r01_bench_1024_summary.txt")).read()[:3000])
