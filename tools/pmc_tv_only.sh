#!/bin/bash
# Raw SQ counters of the tensor-voting launch only (development aid):  bash tools/pmc_tv_only.sh <outdir> [n]
# (VISFD_HIP_TV_FMA in the environment selects the kernel)
out=${1:-gpurun_out/pmc_tv}; n=${2:-1024}
export TMPDIR=/tmp
mkdir -p $out
P1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
i=1
for P in "$P1" "$P2"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/tv_p$i -- python3 tools/pmc_tv.py $n synth > $out/tv_p$i.log 2>&1 || { echo "tv pass $i failed"; tail -5 $out/tv_p$i.log; exit 1; }
  i=$((i+1))
done
python3 tools/pmc_raw_collect.py $out
rm -rf $out/tv_p[0-9]
