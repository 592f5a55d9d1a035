#!/bin/bash
# GPU box, round 5 call v: order of the captured position-only graph (heavy level-1 kernels last vs level order): parity of the
# graphed step, the step, what graph A costs the step (tools/scratch/b_alone_probe.py)
set -u
TAG=${1:-r05v}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout -s KILL 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4 | tee $OUT/pytest_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "heavy kernels last"
M3D_GEO_HEAVY_LAST=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "level order       "
done 2>&1 | tee $OUT/step_geo_order_ab_$TAG.log
{ timeout -s KILL 200 python tools/scratch/b_alone_probe.py; M3D_GEO_HEAVY_LAST=0 timeout -s KILL 200 python tools/scratch/b_alone_probe.py; } 2>&1 | grep "dual-graph" | tee -a $OUT/step_geo_order_ab_$TAG.log
