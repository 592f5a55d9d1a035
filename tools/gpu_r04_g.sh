#!/bin/bash
# GPU box, round 4 call G: the blocks' shortcut branch on a parallel stream (M3D_BRANCH_SHORTCUT) — parity, then the step with
# and without it (graph replay and eager launch); wave priority by phase in the LFA kernels (variant libraries).
set -u
TAG=${1:-r04g}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 500 python -m pytest tests/test_gpu_net.py tests/test_gpu_train.py -m gpu -q --timeout 300 2>&1 | grep -v "^  File\|^Extension modules" | tail -8 > $OUT/pytest_gpu_$TAG.log; tail -4 $OUT/pytest_gpu_$TAG.log | cut -c1-240
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
{
for rep in 1 2; do
  timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph branch=1"
  M3D_BRANCH_SHORTCUT=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph branch=0"
done
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch eager 2>/dev/null | tail -1 | step "eager branch=1"
M3D_BRANCH_SHORTCUT=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch eager 2>/dev/null | tail -1 | step "eager branch=0"
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --precision bf16 2>/dev/null | tail -1 | step "graph bf16 branch=1"
M3D_BRANCH_SHORTCUT=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --precision bf16 2>/dev/null | tail -1 | step "graph bf16 branch=0"
} 2>&1 | tee $OUT/step_branch_ab_$TAG.log
{
  echo "== default"; timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"
  for v in bwdprio1 bwdprio2 fwdprio1 fwdprio2; do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"; done
} > $OUT/lfa_setprio_ab_$TAG.log 2>&1; grep -v amdgpu.ids $OUT/lfa_setprio_ab_$TAG.log
for v in bwdprio1 bwdprio2; do M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph $v"; done | tee -a $OUT/step_branch_ab_$TAG.log
