#!/bin/bash
# GPU box, round 4 call Y: the slot rows of the BatchNorm statistics / backward column sums and the per-column parameters
# loaded all at once (slot_sums) instead of one dependent round trip per slot row, in bn_stats_apply, bn_bwd_apply and the
# dz-on-load GEMM prologue — against the library built from HEAD (variants/libm3d_head.so).
set -u
TAG=${1:-r04y}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_gpu_net.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $OUT/pytest_$TAG.log
{
  echo "== new"; timeout -s KILL 200 python tools/opbench.py bnbwd | grep -v amdgpu
  echo "== head"; M3D_LIB=$V/libm3d_head.so timeout -s KILL 200 python tools/opbench.py bnbwd | grep -v amdgpu
} > $OUT/bnbwd_opbench_$TAG.log 2>&1; grep -E "==|TOTAL|total" $OUT/bnbwd_opbench_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph new"
M3D_LIB=$V/libm3d_head.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph head"
done 2>&1 | tee $OUT/step_$TAG.log
