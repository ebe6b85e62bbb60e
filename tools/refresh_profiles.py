"""Copy the results of tools/profile_round.sh (gpurun_out/<dir>) into profiles/ under a round tag (development aid).

    python tools/refresh_profiles.py gpurun_out/r02 r02
"""
import csv
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src_dir, tag = sys.argv[1:3]
src_dir = os.path.join(ROOT, src_dir)
P = os.path.join(ROOT, "profiles")
stats = os.path.join(src_dir, "kernel_stats.csv")
shutil.copy(stats, os.path.join(P, "%s_bench_1024_kernel_stats.csv" % tag))
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(P, "%s_bench_1024_summary.txt" % tag), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu"
            "   (MI355X, 1024^3, %s)\n" % tag)
    f.write("# 2 timed + 1 warm-up step in the headline (tolerance) mode, 2 + 1 in the other mode, and the extra launches of the\n"
            "#   roofline objects (device copy, Gaussian exact/FMA at 1024^3 and 2048^3, single-axis passes, ridge kernels, tensor voting\n"
            "#   in both modes) + synthetic-input generation\n")
    f.write("# gauss_fused_kernel<5,...>: the roofline launches and the pipeline's plain Gaussians are 8 B/voxel launches (the\n"
            "#   roofline object's ms_per_launch); the second Gaussians of LoG scales also read the minuend (12 B/voxel), and\n"
            "#   the 2048^3 launches take 8x as long, so the average over all calls differs from ms_per_launch\n")
    f.write("%-100s %6s %12s %10s %7s\n" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in rows[:26]:
        f.write("%-100s %6d %12.3f %10.4f %7s\n" % (r["Name"][:100], int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e6, r["Percentage"]))
for a, b in (("bench.json", "%s_bench_1024.json"), ("bench_under_rocprof.json", "%s_bench_1024_under_rocprof.json"),
             ("gauss_traffic.json", "%s_gauss_traffic.json"), ("tv_traffic.json", "%s_tv_traffic.json"),
             ("gauss_fma_traffic.json", "%s_gauss_fma_traffic.json"), ("tv_box_traffic.json", "%s_tv_box_traffic.json"),
             ("gauss_launches.txt", "%s_bench_1024_gauss_launches.txt"), ("pytest_gpu.log", "%s_pytest_gpu.txt")):
    if os.path.exists(os.path.join(src_dir, a)):
        shutil.copy(os.path.join(src_dir, a), os.path.join(P, b % tag))
print(open(os.path.join(P, "%s_bench_1024_summary.txt" % tag)).read()[:4000])
