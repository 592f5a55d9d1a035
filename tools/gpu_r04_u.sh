#!/bin/bash
# GPU box, round 4 call U: weight-gradient batch — steps in flight and wave budget of the 16-tile waves, inside the step.
set -u
TAG=${1:-r04u}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"
for v in wd4 wd1 wb3072 wb6144 wb2048; do M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph $v"; done
done 2>&1 | tee $OUT/step_$TAG.log
