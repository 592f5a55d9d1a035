#!/bin/bash
# wavefront-cooperative self-kNN: bit-exact parity, then per-level A/B against the per-lane kernels
set -u
TAG=${1:-r02r}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_coop_$TAG.log; : > $L
timeout -s KILL 600 python -m pytest tests -m gpu -q -x -k "knn or eval_logits or full_size or seeded or golden or interpolate or lookahead or dense or reference_sizes" 2>&1 | tail -5 >> $L
for c in 1 0 1 0; do
  echo "=== M3D_KNN_COOP=$c" >> $L
  M3D_KNN_COOP=$c timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
done
for v in myria3d_amd/variants/libm3d_knnc_*.so; do
  [ -f "$v" ] || continue
  M3D_LIB=$PWD/$v timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
done
for c in 1 0; do
  echo "=== bench M3D_KNN_COOP=$c" >> $L
  M3D_KNN_COOP=$c timeout 300 python bench.py --steps 40 --warmup 10 --skip-cpu-baseline --skip-roofline --skip-extras 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'])" >> $L 2>&1
  M3D_KNN_COOP=$c timeout 300 python bench.py --steps 40 --warmup 10 --no-lookahead --skip-cpu-baseline --skip-roofline --skip-extras 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-lookahead', d['ms_per_step'], d['fwd_only']['ms_per_step'])" >> $L 2>&1
done
grep -v amdgpu.ids $L
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
bash tools/gpu_pmc.sh ${TAG}_q1 "$C" python tools/knn_bench.py pmc | grep -i knn_query
