#!/bin/bash
set -u
TAG=${1:-r02f}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 300 python -m pytest tests -m gpu -q -x -k "tile_select or forward_like or wgrad or shared_layer or residual or train_forward_backward or train_steps or gemm" 2>&1 | tail -8
timeout -s KILL 200 python tools/opbench.py wgrad 2>&1 | tail -32 > $OUT/opbench_wgrad_$TAG.log; cat $OUT/opbench_wgrad_$TAG.log | cut -c1-40,100-180
for i in 1 2; do
timeout -s KILL 200 python bench.py --lookahead --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}_look.err | tail -1 | cut -c1-300
done
bash tools/gpu_trace_analyze.sh ${TAG} --lookahead 2>&1 | tail -45
