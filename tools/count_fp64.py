"""FP64 vector instructions of the eigen kernels, counted in the compiled code (static count of the kernels' straight-line
common path plus its rare branches; FMA forms count 2 flop, everything else -- add, mul, conversions, min/max -- 1):

    python tools/count_fp64.py          # compiles csrc/ridge.hip for gfx950 to assembly and prints per-kernel counts

bench.py's `roofline_ridge` prices the kernels' FP64 work with these numbers (FP64_FLOP below)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# per voxel the kernel works on (ridge_directions: per voxel above the threshold); round 3 build
FP64_FLOP = {"ridge_score_kernel": 153, "ridge_directions_kernel": 987, "tensor_saliency_kernel": 146}


def main():
    out = os.path.join(tempfile.mkdtemp(), "ridge.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "visfd_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                           "-S", "--cuda-device-only", os.path.join(ROOT, "visfd_amd", "csrc", "ridge.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    name, stats = None, {}
    for line in open(out):
        m = re.match(r"^_ZN2vh12_GLOBAL__N_1\d+([a-z_0-9]+?_kernel)E\S*:", line)
        if m:
            name = m.group(1)
            stats[name] = [0, 0, 0]
            continue
        m = re.match(r"^\s+(v_\S+)", line)
        if m and name:
            stats[name][0] += 1
            if "f64" in m.group(1):
                stats[name][1] += 1
                if "fma" in m.group(1):
                    stats[name][2] += 1
    for k, (valu, f64, fma) in sorted(stats.items()):
        print("%-28s VALU %5d   FP64 %4d (FMA %3d)   FP64 flop %5d" % (k, valu, f64, fma, f64 + fma))
    bad = [k for k, v in FP64_FLOP.items() if k in stats and stats[k][1] + stats[k][2] != v]
    if bad:
        print("FP64_FLOP is stale for: " + ", ".join(bad))
        sys.exit(1)


if __name__ == "__main__":
    main()
