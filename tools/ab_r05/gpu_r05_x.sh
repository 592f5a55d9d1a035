#!/bin/bash
# GPU box, round 5 call x: weight gradients of the row-stream SharedMLP layers inside their input-gradient launches
# (m3d_bn_dgrad_wgrad_f32) vs the weight-gradient batches (M3D_WGRAD_IN_DGRAD=0): parity, the step, kernel times
set -u
TAG=${1:-r05x}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout -s KILL 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_train.py -x -q -m gpu -k "train or grad or replay or graph or shared or dgrad or ddp or parallel" 2>&1 | grep -E "passed|failed|rror" | tail -4 | tee $OUT/pytest_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "weight gradients inside dgrad"
M3D_WGRAD_IN_DGRAD=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "weight-gradient batches      "
done 2>&1 | tee $OUT/step_wgrad_in_dgrad_ab_$TAG.log
bash tools/gpu_trace_analyze.sh $TAG > $OUT/trace_$TAG.log 2>&1
grep -E "wgrad|gemm_rowstream" $OUT/step_timeline_$TAG.csv | tail -40
