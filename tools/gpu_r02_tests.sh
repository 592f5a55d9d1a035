#!/bin/bash
# GPU box, round 2: full parity suite (incl. the opt-in lookahead test), the default bench line, a graph-mode trace
# analysis of one training step.  usage: tools/gpu_r02_tests.sh TAG
set -u
TAG=${1:-r02a}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
M3D_EXPERIMENTAL=1 timeout -s KILL 900 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40 > $OUT/pytest_gpu_$TAG.log
tail -25 $OUT/pytest_gpu_$TAG.log
timeout -s KILL 400 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
timeout -s KILL 120 python bench.py --lookahead --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}_look.err | tail -1 > $OUT/bench_${TAG}_look.json
cat $OUT/bench_${TAG}_look.json; tail -3 $OUT/bench_${TAG}_look.err
bash tools/gpu_trace_analyze.sh $TAG 2>&1 | tail -60 > $OUT/trace_analysis_$TAG.log
cat $OUT/trace_analysis_$TAG.log
