#!/bin/bash
# GPU box, round 4 call P: LFA partial-sum reduce with one writer per dW element (no atomics) — parity, the step, timeline.
set -u
TAG=${1:-r04p}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error|Error|assert|FAILED" | tail -12 > $OUT/pytest_gpu_$TAG.log; cat $OUT/pytest_gpu_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2 3; do timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"; M3D_LIB=$GRAFT_REPO_ROOT/myria3d_amd/variants/libm3d_redold.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph redold"; done 2>&1 | tee $OUT/step_$TAG.log
bash tools/gpu_trace_analyze.sh $TAG > $OUT/trace_analyze_$TAG.log 2>&1; grep -E "^step:|per queue|main queue" $OUT/trace_analyze_$TAG.log
grep -E "lfa_bwd_reduce|wgrad_reduce|colsum|adam|ce_" $OUT/step_timeline_$TAG.csv
