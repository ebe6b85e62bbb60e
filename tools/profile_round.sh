#!/bin/bash
# The measurements a round's profiles/ entries come from; run on the GPU box from the repo root:
#   bash tools/profile_round.sh gpurun_out/r02 [n]
# then, back in the container:  python tools/refresh_profiles.py gpurun_out/r02 r02
# Steps are joined so that nothing runs after a failed GPU step.
out=${1:-gpurun_out/round}; n=${2:-1024}
export TMPDIR=/tmp
mkdir -p $out
set -e
python3 bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu > $out/bench_under_rocprof.json 2> $out/prof.err
cp $out/prof/*/*kernel_stats.csv $out/kernel_stats.csv
# the Gaussian launches of that run by grid size (1024^3 and 2048^3 launches share one kernel name in the statistics)
python3 - $out/prof > $out/gauss_launches.txt <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0])))
groups = {}
for r in rows:
    if "gauss_fused_kernel<5" not in r["Kernel_Name"]:   # (exact and FMA instantiations together)
        continue
    key = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("# gauss_fused_kernel<5,...> launches of the rocprofv3 --kernel-trace run of bench.py, grouped by grid size (ms)")
for k, v in sorted(groups.items()):
    v.sort()
    print("grid %s: %d launches, mean %.4f  median %.4f  min %.4f  max %.4f" % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[0], v[-1]))
PY
rm -rf $out/prof
# HBM traffic per launch from PMC counters, each counter in a pass of its own (MI355X_MICROARCH.md): the exact and the
# tolerance form of the Gaussian and of tensor voting
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/g_$c -- python3 tools/pmc_traffic.py $n > $out/g_$c.log 2>&1
  VISFD_HIP_GAUSS_FMA=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/gf_$c -- python3 tools/pmc_traffic.py $n > $out/gf_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/t_$c -- python3 tools/pmc_tv.py $n synth > $out/t_$c.log 2>&1
  VISFD_HIP_TV_FMA=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/tp_$c -- python3 tools/pmc_tv.py $n synth > $out/tp_$c.log 2>&1
done
python3 tools/pmc_traffic_collect.py $out/g_FETCH_SIZE $out/g_WRITE_SIZE gauss_fused_kernel $out/gauss_traffic.json 8 $n "sigma=2 h=5, exact"
python3 tools/pmc_traffic_collect.py $out/gf_FETCH_SIZE $out/gf_WRITE_SIZE gauss_fused_kernel $out/gauss_fma_traffic.json 8 $n "sigma=2 h=5, tolerance mode (gauss_fma)"
python3 tools/pmc_traffic_collect.py $out/t_FETCH_SIZE $out/t_WRITE_SIZE tv_boxx_kernel $out/tv_traffic.json 40 $n "sigma_tv=8.66 h=12, 5 % salient, bench synthetic volume, exact kernel (exact form of tv_box.hip)"
python3 tools/pmc_traffic_collect.py $out/tp_FETCH_SIZE $out/tp_WRITE_SIZE tv_box_kernel $out/tv_box_traffic.json 40 $n "sigma_tv=8.66 h=12, 5 % salient, bench synthetic volume, tolerance mode (tv_fma)"
rm -rf $out/g_FETCH_SIZE $out/g_WRITE_SIZE $out/gf_FETCH_SIZE $out/gf_WRITE_SIZE $out/t_FETCH_SIZE $out/t_WRITE_SIZE $out/tp_FETCH_SIZE $out/tp_WRITE_SIZE
