#!/bin/bash
# GPU box, round 5 call n: same-box A/B of the in-lane neighbourhood layout in the LFA backward (ch >= 64): per-level times,
# the step, and the dominant kernel timed inside training steps, product vs -DLFA_BWD_INL=0
set -u
TAG=${1:-r05n}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
V=$ROOT/myria3d_amd/variants/libm3d_noinl.so
{ echo "== in-lane (product)"; timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa" | sed "s/fwd.*bwd/bwd/"
  echo "== cross-lane softmax (LFA_BWD_INL=0)"; M3D_LIB=$V timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa" | sed "s/fwd.*bwd/bwd/"; } 2>&1 | grep -v amdgpu.ids | tee $OUT/lfa_bwd_inl_ab_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['ms_per_step'], 'ms; dominant kernel in-step', r['avg_launch_ms'], 'frac', r['frac'], 'isolated', r['isolated_launch_ms'])"; }
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "in-lane   "
M3D_LIB=$V timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "cross-lane"
done 2>&1 | tee $OUT/step_inl_ab_$TAG.log
