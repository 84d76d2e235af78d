#!/bin/bash
# Kernel trace of 10 steps of BASELINE configs[1] (att2in2 MLE, B = 64) in the bf16 variant -> per-kernel step breakdown.
# usage: bash tools/quick_profile_c2.sh <tag>
set -e
tag=${1:-qc2}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $R/tools/config_bench.py 10 c2bf16 > $out/bench.log 2> $out/trace.err
cd $R
kt=$(find $out/trace -name '*kernel_trace.csv' | head -1)
python tools/trace_summary.py $kt 10 $out/step_breakdown.md $out/step_sequence.txt 1 > $out/trace_summary.log
rm -rf $out/trace
