#!/bin/bash
# GPU box, round 4 call H: wave priority by phase in the LFA kernels (phase masks / levels), software-pipelined row-stream
# GEMMs (next tile's fragments in flight during this tile's MFMAs / stores) — per kernel, parity, and inside the step.
set -u
TAG=${1:-r04h}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
{
timeout -s KILL 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -k "lfa" 2>&1 | grep -E "passed|failed|error" | tail -2
M3D_LIB=$V/libm3d_rspf3.so timeout -s KILL 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --timeout 300 -k "gemm or shared or dgrad or train or golden or reference" 2>&1 | grep -E "passed|failed|error" | tail -3
} > $OUT/pytest_gpu_$TAG.log 2>&1; cat $OUT/pytest_gpu_$TAG.log
{
  echo "== default (backward mask 21)"; timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"
  for v in bp0 bp21l3 bp5 bp20 bp17 bp16 bp4 bp1 bp29 fp5 fp1 fp4l3; do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"; done
} > $OUT/lfa_setprio_masks_$TAG.log 2>&1; grep -v amdgpu.ids $OUT/lfa_setprio_masks_$TAG.log | grep -E "==|level 1|level 2 ch= 64|level 4 ch=256"
{
  echo "== default"; timeout -s KILL 300 python tools/opbench.py gemm bnbwd
  for v in rspf1 rspf3; do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 300 python tools/opbench.py gemm bnbwd; done
} > $OUT/gemm_rowstream_prefetch_$TAG.log 2>&1; grep -E "==|TOTAL|M=204800|M= 51200" $OUT/gemm_rowstream_prefetch_$TAG.log | cut -c1-200
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
{
for rep in 1 2; do timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"; done
for v in bp0 bp21l3 rspf1 rspf3; do M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph $v"; done
} 2>&1 | tee $OUT/step_$TAG.log
