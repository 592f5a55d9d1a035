#!/bin/bash
# GPU box, round 4 call K: full parity suite; when the two graphs of a step run relative to each other (no tracer); the step.
set -u
TAG=${1:-r04k}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $OUT/pytest_gpu_$TAG.log; cat $OUT/pytest_gpu_$TAG.log
timeout -s KILL 300 python tools/graph_overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/graph_overlap_probe_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"; done | tee $OUT/step_$TAG.log
