#!/bin/bash
# Raw SQ counters and HBM bytes of the blob stage's kernels (development aid):  bash tools/pmc_blob.sh <outdir> [n]
out=${1:-gpurun_out/pmc_blob}; n=${2:-1024}
export TMPDIR=/tmp
mkdir -p $out
P1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/blob_p$i -- python3 tools/pmc_blob.py $n 4 3 > $out/blob_p$i.log 2>&1 || { echo "blob pass $i failed"; tail -5 $out/blob_p$i.log; exit 1; }
  i=$((i+1))
done
python3 tools/pmc_raw_collect.py $out
rm -rf $out/blob_p[0-9]
