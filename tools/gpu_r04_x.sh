#!/bin/bash
# GPU box, round 4 call X: deferred weight-gradient batches / LFA partial-sum reduces launched EARLY on the gradient side
# stream, next to the long LFA backward kernels (M3D_WGRAD_EARLY bits: 1 first LFA, 2 every LFA, 4 LFA reduces per level).
# (the switch lost and was removed from ops.py after this run; kept for the record of profiles/r04x_*)
set -u
TAG=${1:-r04x}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
M3D_WGRAD_EARLY=6 timeout -s KILL 400 python -m pytest tests/test_gpu_train.py tests/test_gpu_net.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_early_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eager', d.get('eager_ms_per_step'), 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do
for m in 0 1 2 4 5 6; do M3D_WGRAD_EARLY=$m timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "early=$m"; done
done 2>&1 | tee $OUT/step_$TAG.log
