#!/bin/bash
# GPU box, round 5 call r: lanes per list of the long-list gather (4 vs 8): test, step, kernel times
set -u
TAG=${1:-r05r}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
V=$ROOT/myria3d_amd/variants/libm3d_lpl8.so
for lib in "" $V; do M3D_LIB=$lib timeout -s KILL 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "reverse or edge_rows" 2>&1 | tail -2; done | tee $OUT/pytest_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "4 lanes per list"
M3D_LIB=$V timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "8 lanes per list"
done 2>&1 | tee $OUT/step_lpl_ab_$TAG.log
M3D_LIB=$V bash tools/gpu_trace_analyze.sh $TAG > $OUT/trace_$TAG.log 2>&1
grep -E "gather_sum_rows4|lfa_bwd_small|rev_|bn_bwd_reduce" $OUT/step_timeline_$TAG.csv | tail -12
