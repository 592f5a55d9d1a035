#!/bin/bash
# GPU box, round 4 call F: LFA tile ownership (rows first) — parity of the new default, per-kernel timing of the variants, the step.
set -u
TAG=${1:-r04f}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --timeout 300 -k "lfa or train or golden or bf16 or reference" 2>&1 | grep -v "^  File\|^Extension modules" | tail -8 > $OUT/pytest_gpu_$TAG.log; tail -4 $OUT/pytest_gpu_$TAG.log | cut -c1-240
{
  echo "== default (rows first, square dW blocks)"; timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"
  for v in bwd_old bwd_minw3 bwd_rowonly bwd_sqonly bwd_bpre fwd_old fwd_bpre; do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"; done
} > $OUT/lfa_tile_ownership_ab_$TAG.log 2>&1; grep -v amdgpu.ids $OUT/lfa_tile_ownership_ab_$TAG.log
for rep in 1 2; do timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done | tee $OUT/step_$TAG.log
M3D_LIB=$V/libm3d_bwd_old.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step (bwd_old)', d['ms_per_step'], 'ms; roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $OUT/step_$TAG.log
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph --precision bf16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step bf16', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'])" | tee -a $OUT/step_$TAG.log
