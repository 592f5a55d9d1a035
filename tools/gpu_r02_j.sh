#!/bin/bash
set -u
TAG=${1:-r02j}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
for i in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}.err | tail -1 | cut -c1-250
done
tail -3 $OUT/bench_${TAG}.err
bash tools/gpu_trace_analyze.sh ${TAG} 2>&1 | head -24
