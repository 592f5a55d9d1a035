#!/bin/bash
set -u
TAG=${1:-r02g}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 600 python -m pytest tests -m gpu -q -x -k "lookahead or seeded or reference_shape or full_size" 2>&1 | tail -4
for i in 1 2; do
timeout -s KILL 200 python bench.py --no-lookahead --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}_base.err | tail -1 | cut -c1-300
timeout -s KILL 200 python bench.py --lookahead --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}_look.err | tail -1 | cut -c1-300
done
tail -3 $OUT/bench_${TAG}_look.err
bash tools/gpu_trace_analyze.sh ${TAG} --lookahead 2>&1 | tail -30
