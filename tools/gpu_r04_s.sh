#!/bin/bash
# GPU box, round 4 call S: dropout mask indexed by the caller's row (reproducible across runs), input permutation inside fc0's
# GEMM — full parity suite, the step.
set -u
TAG=${1:-r04s}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error|Error|assert|FAILED" | tail -12 > $OUT/pytest_gpu_$TAG.log; cat $OUT/pytest_gpu_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2 3; do timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"; done 2>&1 | tee $OUT/step_$TAG.log
