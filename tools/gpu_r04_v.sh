#!/bin/bash
# GPU box, round 4 call V: wave count of the k-loop GEMMs (GEMM_KL_MINWAVES), per layer and inside the step.
set -u
TAG=${1:-r04v}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
{
  echo "== default (1536)"; timeout -s KILL 300 python tools/opbench.py gemm | grep -v amdgpu | grep -E "M= 12800|M=  3200|M=   800|TOTAL"
  for v in kl1024 kl3072 kl6144; do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 300 python tools/opbench.py gemm | grep -v amdgpu | grep -E "M= 12800|M=  3200|M=   800|TOTAL"; done
} > $OUT/gemm_kl_minwaves_$TAG.log 2>&1; grep -E "==|TOTAL" $OUT/gemm_kl_minwaves_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"
for v in kl1024 kl3072 kl6144; do M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph $v"; done
done 2>&1 | tee $OUT/step_$TAG.log
