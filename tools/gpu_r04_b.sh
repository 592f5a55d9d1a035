#!/bin/bash
# GPU box, round 4 call B: parity suite file by file (a crash in one file does not hide the others), XCD-aware order A/B
# (LFA forward / backward, row scatter-add; per kernel and inside the step), kNN drain A/B, FPS, the N > 1 launch-form probe.
set -u
TAG=${1:-r04b}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
: > $OUT/pytest_gpu_$TAG.log
for f in tests/test_gpu_ops.py tests/test_gpu_pointnet2.py tests/test_gpu_net.py tests/test_gpu_train.py tests/test_gpu_prep.py tests/test_tiling.py; do
  echo "=== $f" >> $OUT/pytest_gpu_$TAG.log
  timeout -s KILL 400 python -m pytest $f -m gpu -q --timeout 300 -x 2>&1 | grep -v "^  File\|^Extension modules" | tail -25 >> $OUT/pytest_gpu_$TAG.log
done
grep -E "^===|passed|failed|error|Error|Fatal|core" $OUT/pytest_gpu_$TAG.log | cut -c1-220
{
  timeout -s KILL 120 python tools/knn_bench.py
  for v in knn_nonet knn_net16only knn_net12; do M3D_LIB=$V/libm3d_$v.so timeout -s KILL 120 python tools/knn_bench.py; done
} 2>&1 | grep -E "knn_bench|Error|error" > $OUT/knn_ab_$TAG.log; cat $OUT/knn_ab_$TAG.log
{
  echo "== default (XCD-aware order)"; timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"
  echo "== round-3 order, LFA backward"; M3D_LIB=$V/libm3d_noxcd_bwd.so timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"
  echo "== round-3 order, LFA forward"; M3D_LIB=$V/libm3d_noxcd_fwd.so timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level"
} > $OUT/lfa_xcd_order_ab_$TAG.log 2>&1; grep -v amdgpu.ids $OUT/lfa_xcd_order_ab_$TAG.log
{
  for lib in default noxcd_all; do
    for rep in 1 2; do
      if [ $lib = default ]; then L=""; else L="$V/libm3d_$lib.so"; fi
      M3D_LIB=$L timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --skip-roofline --launch graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step lib=$lib', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'])"
    done
  done
} > $OUT/step_xcd_ab_$TAG.log 2>&1; cat $OUT/step_xcd_ab_$TAG.log
for c in none noopt pg_gloo pg eager captured; do timeout -s KILL 150 python tools/collective_probe.py $c 2>&1 | grep collective_probe; done > $OUT/collective_probe_$TAG.log; cat $OUT/collective_probe_$TAG.log
timeout -s KILL 200 python bench.py --mode pointnet2 --steps 3 --tiles 16 --points 40000 --neighbors 32 2>/dev/null | tail -1 > $OUT/pointnet2_$TAG.json; cut -c1-400 $OUT/pointnet2_$TAG.json
