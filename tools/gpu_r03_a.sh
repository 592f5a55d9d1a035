#!/bin/bash
# GPU box, round 3 first call: full parity suite (new: GraphedStep, reference fixture, flattened-path gradients, 1-rank RCCL),
# the default bench line with the new legs, a graph-mode step timeline.  usage: tools/gpu_r03_a.sh TAG
set -u
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -40 > $OUT/pytest_gpu_$TAG.log; tail -25 $OUT/pytest_gpu_$TAG.log
timeout -s KILL 900 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
cut -c1-3000 $OUT/bench_$TAG.json; tail -5 $OUT/bench_$TAG.err
bash tools/gpu_trace_analyze.sh $TAG 2>&1 | tail -40
