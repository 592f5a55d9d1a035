#!/bin/bash
# GPU box, round 5 call u: edge rows stored in reverse-list order (contiguous per point; the gather streams them) vs in edge
# order (gather through the index table): parity, step, kernel times
set -u
TAG=${1:-r05u}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout -s KILL 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_train.py -x -q -m gpu -k "lfa or train or grad or replay or graph or reverse or csr" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/pytest_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "rows in list order"
M3D_LFA_EDGE_SLOTS=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "rows in edge order"
done 2>&1 | tee $OUT/step_edge_slots_ab_$TAG.log
bash tools/gpu_trace_analyze.sh $TAG > $OUT/trace_$TAG.log 2>&1
grep -E "gather_sum_rows4|lfa_bwd_small|rev_" $OUT/step_timeline_$TAG.csv | tail -12
