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
rm -rf $out/prof
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/g_$c -- python3 tools/pmc_traffic.py $n > $out/g_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/t_$c -- python3 tools/pmc_tv.py $n synth > $out/t_$c.log 2>&1
done
python3 tools/pmc_traffic_collect.py $out/g_FETCH_SIZE $out/g_WRITE_SIZE gauss_fused_kernel $out/gauss_traffic.json 8 $n "sigma=2 h=5"
python3 tools/pmc_traffic_collect.py $out/t_FETCH_SIZE $out/t_WRITE_SIZE tv_tiled_kernel $out/tv_traffic.json 40 $n "sigma_tv=8.66 h=12, 5 % salient, bench synthetic volume"
rm -rf $out/g_FETCH_SIZE $out/g_WRITE_SIZE $out/t_FETCH_SIZE $out/t_WRITE_SIZE
