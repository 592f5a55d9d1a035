#!/bin/bash
# GPU box, round 5 call o: step A/B — in-lane backward layout from ch = 128 (product) / from ch = 64 / nowhere
set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['ms_per_step'], 'ms; dominant kernel in-step', r['avg_launch_ms'], 'frac', r['frac'], 'isolated', r['isolated_launch_ms'])"; }
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 200 2>/dev/null | tail -1 | step "in-lane from ch=128 (product)"
M3D_LIB=$ROOT/myria3d_amd/variants/libm3d_inl64.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 200 2>/dev/null | tail -1 | step "in-lane from ch=64           "
M3D_LIB=$ROOT/myria3d_amd/variants/libm3d_noinl.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 200 2>/dev/null | tail -1 | step "cross-lane everywhere         "
done 2>&1 | tee $OUT/step_inl_minch_ab_r05o.log
