#!/bin/bash
# GPU box, round 4 call Y3: straight-line row loops in the BatchNorm streaming kernels (bn_bwd_reduce: eight rows of dy / z in
# flight, column constants in front of the loop; bn_stats_apply, bn_bwd_apply: four float4 per trip) — whole GPU suite, the
# per-layer column-sum / fused dgrad times, then the step against HEAD~1's library (variants/libm3d_head.so = cf4159b..8771c0e).
set -u
TAG=${1:-r04y3}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $OUT/pytest_$TAG.log
{
  echo "== new"; timeout -s KILL 200 python tools/opbench.py bnbwd bn | grep -v amdgpu
  echo "== head"; M3D_LIB=$V/libm3d_head.so timeout -s KILL 200 python tools/opbench.py bnbwd bn | grep -v amdgpu
} > $OUT/bnbwd_opbench_$TAG.log 2>&1; grep -E "==|TOTAL|^bn M" $OUT/bnbwd_opbench_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph new"
M3D_LIB=$V/libm3d_head.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph head"
done 2>&1 | tee $OUT/step_$TAG.log
