#!/bin/bash
# L2 counters of the bench step (one PMC pass, counters + kernel trace only).  usage: bash tools/measure_pmc_l2.sh <tag>
set -e
tag=${1:-l2}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_l2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/pmc_l2.log 2>&1
cd $R
cc=$(find $out/pmc_l2 -name '*counter_collection.csv' | head -1)
python tools/pmc_l2.py $cc $out/pmc_l2.json > $out/pmc_l2_summary.log
rm -rf $out/pmc_l2
head -16 $out/pmc_l2_summary.log
