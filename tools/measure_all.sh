#!/bin/bash
# Full measurement set for profiles/: GPU tests, bench line, kernel trace + stats, the PMC passes (FETCH_SIZE, WRITE_SIZE,
# MFMA utilisation: three separate counter-only runs).
# Run on the GPU box from the repo root: bash tools/measure_all.sh <tag>
set -e
tag=${1:-m}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $out/tests.log 2>&1
python bench.py --steps 30 --warmup 5 > $out/bench.json 2> $out/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > $out/bench_rocprof.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $out/pmc_mfma.log 2>&1
cd $R
kt=$(find $out/trace -name '*kernel_trace.csv' | head -1)
ks=$(find $out/trace -name '*kernel_stats.csv' | head -1)
python tools/trace_summary.py $kt 10 $out/step_breakdown.md $out/step_sequence.txt > $out/trace_summary.log
cp $ks $out/kernel_stats.csv
fc=$(find $out/pmc_fetch -name '*counter_collection.csv' | head -1)
wc=$(find $out/pmc_write -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py $fc $wc $out/pmc_traffic.json > $out/pmc_summary.log
mc=$(find $out/pmc_mfma -name '*counter_collection.csv' | head -1)
python tools/pmc_mfma.py $mc $out/pmc_mfma.json > $out/pmc_mfma_summary.log
# the raw traces are large: keep only the summaries
rm -rf $out/trace $out/pmc_fetch $out/pmc_write $out/pmc_mfma
tail -3 $out/tests.log; cat $out/bench.json
