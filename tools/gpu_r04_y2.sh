#!/bin/bash
# GPU box, round 4 call Y2: call Y's slot_sums + float4 column sums for the bias gradients (colsum4_kernel) + the register path
# of the cross-entropy forward (four rows per thread, all loads up front) — whole GPU suite, then the step against HEAD's library.
set -u
TAG=${1:-r04y2}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $OUT/pytest_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph new"
M3D_LIB=$V/libm3d_head.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph head"
done 2>&1 | tee $OUT/step_$TAG.log
