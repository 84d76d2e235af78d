#!/bin/bash
# MFMA-utilisation counters of the bench step (one PMC pass, counters + kernel trace only).  usage: bash tools/measure_pmc_mfma.sh <tag>
set -e
tag=${1:-mfma}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/pmc_mfma.log 2>&1
cd $R
cc=$(find $out/pmc_mfma -name '*counter_collection.csv' | head -1)
head -3 $cc > $out/pmc_mfma_head.txt
python tools/pmc_mfma.py $cc $out/pmc_mfma.json > $out/pmc_mfma_summary.log
rm -rf $out/pmc_mfma
head -12 $out/pmc_mfma_summary.log
