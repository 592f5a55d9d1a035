#!/bin/bash
# GPU box, round 5 call p: the wave-autonomous LFA backward at ch = 8 / 16 (lfa_bwd_small_kernel): parity, per-level times and
# the step against -DLFA_BWD_SMALL=0 (the four-wave tile kernel), 4 vs 3 waves per SIMD, and without the dx atomics
set -u
TAG=${1:-r05p}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
VD=$ROOT/myria3d_amd/variants
timeout -s KILL 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -x -q -m gpu -k "lfa or train or grad" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $OUT/pytest_lfa_$TAG.log
{ for v in product small4 nosmall smallnoatom; do
    echo "== $v"
    if [ $v = product ]; then timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa" | sed "s/fwd.*bwd/bwd/" | head -3
    else M3D_LIB=$VD/libm3d_$v.so timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa" | sed "s/fwd.*bwd/bwd/" | head -3; fi
  done; } 2>&1 | grep -v amdgpu.ids | tee $OUT/lfa_bwd_small_ab_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "product (small, 3 waves)"
M3D_LIB=$VD/libm3d_small4.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "small, 4 waves (spills)  "
M3D_LIB=$VD/libm3d_nosmall.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --steps 100 2>/dev/null | tail -1 | step "tile kernel (round 4 form)"
done 2>&1 | tee $OUT/step_small_ab_$TAG.log
