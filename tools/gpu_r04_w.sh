#!/bin/bash
# GPU box, round 4 call W: LFA loads in two dependent round trips (centre position, weight fragments and neighbour ids first;
# then neighbour rows and positions together) against the five-round-trip kernels of r04zzz. Parity, per-op time, step.
set -u
TAG=${1:-r04w}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -x -q -k "lfa or train or golden or reference or parity" 2>&1 | tail -3 | tee $OUT/pytest_lfa_$TAG.log
{
  echo "== new"; timeout -s KILL 200 python tools/opbench.py lfa | grep -v amdgpu
  echo "== fwdnew (old bwd)"; M3D_LIB=$V/libm3d_fwdnew.so timeout -s KILL 200 python tools/opbench.py lfa | grep -v amdgpu
  echo "== old"; M3D_LIB=$V/libm3d_lfaold.so timeout -s KILL 200 python tools/opbench.py lfa | grep -v amdgpu
} > $OUT/lfa_opbench_$TAG.log 2>&1; cat $OUT/lfa_opbench_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph new"
for v in fwdnew lfaold; do M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph $v"; done
done 2>&1 | tee $OUT/step_$TAG.log
