#!/bin/bash
# GPU box, round 4 call I: row-stream GEMMs with a batch of tiles per wave trip (GEMM_RS_TB), fused step prologue / loss
# finalize / Adam tick (ABI 14) — parity, per kernel, inside the step; backward priority masks once more.
set -u
TAG=${1:-r04i}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $OUT/pytest_gpu_$TAG.log; cat $OUT/pytest_gpu_$TAG.log
{
  echo "== default"; timeout -s KILL 300 python tools/opbench.py gemm bnbwd
  for v in rstb1 rstb2; do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 300 python tools/opbench.py gemm bnbwd; done
} > $OUT/gemm_rowstream_batch_$TAG.log 2>&1; grep -E "==|TOTAL" $OUT/gemm_rowstream_batch_$TAG.log | cut -c1-200
{
  for rep in 1 2; do
  echo "== default (backward mask 21)"; timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"
  for v in bp0 bp17 bp1 bp20; do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"; done
  done
} > $OUT/lfa_setprio_masks_$TAG.log 2>&1
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
{
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph default"
for v in rstb1 bp0 bp17 bp1; do M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph $v"; done
done
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch eager 2>/dev/null | tail -1 | step "eager default"
} 2>&1 | tee $OUT/step_$TAG.log
bash tools/gpu_trace_analyze.sh $TAG > $OUT/trace_analyze_$TAG.log 2>&1; tail -30 $OUT/trace_analyze_$TAG.log
