#!/bin/bash
# GPU box, round 4 call T: software-pipelined loads in the weight-gradient kernels (WGRAD_PIPE) — per layer and inside the step.
set -u
TAG=${1:-r04t}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
{
  echo "== default"; timeout -s KILL 300 python tools/opbench.py wgrad | grep -v amdgpu
  for v in wp1 wp2 wp3; do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 300 python tools/opbench.py wgrad | grep -v amdgpu; done
} > $OUT/wgrad_pipe_$TAG.log 2>&1; grep -E "==|TOTAL" $OUT/wgrad_pipe_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"
for v in wp1 wp2 wp3; do M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph $v"; done
done 2>&1 | tee $OUT/step_$TAG.log
M3D_LIB=$V/libm3d_wp3.so timeout -s KILL 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --timeout 300 -k "wgrad or train or shared or golden" 2>&1 | grep -E "passed|failed" | tail -2
