#!/bin/bash
# Kernel trace of 10 bench steps of the bf16 variant -> per-kernel step breakdown.  usage: bash tools/quick_profile_bf16.sh <tag>
set -e
tag=${1:-qbf}
R=$(pwd)
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --compute-dtype bf16 > $out/bench_rocprof.json 2> $out/trace.err
cd $R
kt=$(find $out/trace -name '*kernel_trace.csv' | head -1)
python tools/trace_summary.py $kt 10 $out/step_breakdown.md $out/step_sequence.txt > $out/trace_summary.log
rm -rf $out/trace
