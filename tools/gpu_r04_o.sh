#!/bin/bash
# GPU box, round 4 call O: row-stream GEMMs with a batch of tiles per wave trip, branch-free (GEMM_RS_TB) — parity, per
# kernel against the one-tile-per-trip build, inside the step.
set -u
TAG=${1:-r04o}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error|Error|assert|FAILED" | tail -12 > $OUT/pytest_gpu_$TAG.log; cat $OUT/pytest_gpu_$TAG.log
{
  echo "== default (batched)"; timeout -s KILL 300 python tools/opbench.py gemm bnbwd
  echo "== rstb1"; M3D_LIB=$V/libm3d_rstb1.so timeout -s KILL 300 python tools/opbench.py gemm bnbwd
} > $OUT/gemm_rowstream_batch_$TAG.log 2>&1; grep -E "==|TOTAL|M=204800" $OUT/gemm_rowstream_batch_$TAG.log | cut -c1-190
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
{
for rep in 1 2 3; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"
M3D_LIB=$V/libm3d_rstb1.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph rstb1"
done
} 2>&1 | tee $OUT/step_$TAG.log
