#!/bin/bash
# GPU box, round 5 call q: dx of the 8 / 16-channel LFA layers through per-edge rows + reverse neighbour lists (no atomics):
# parity, then the step with and without (M3D_LFA_EDGE_ROWS=0), and the kernel trace of one step
set -u
TAG=${1:-r05q}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout -s KILL 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_train.py -x -q -m gpu -k "lfa or train or grad or replay or graph or reverse or csr" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $OUT/pytest_lfa_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "edge rows + reverse lists"
M3D_LFA_EDGE_ROWS=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "float atomics            "
done 2>&1 | tee $OUT/step_edge_rows_ab_$TAG.log
bash tools/gpu_trace_analyze.sh $TAG > $OUT/trace_$TAG.log 2>&1
grep -E "lfa_bwd|gather_sum|csr_" $OUT/step_timeline_$TAG.csv | head -30
