#!/bin/bash
# cell-size sweep for the cooperative self-kNN (M3D_KNN_CELL_TARGET = points per grid column) + SQ counters at two settings
set -u
TAG=${1:-r02s}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_coop_cells_$TAG.log; : > $L
timeout -s KILL 600 python -m pytest tests -m gpu -q -x -k "knn or eval_logits or full_size or golden or dense" 2>&1 | tail -3 >> $L
for t in ${TARGETS:-7 10 14 20 28 40}; do
  echo "=== M3D_KNN_CELL_TARGET=$t coop" >> $L
  M3D_KNN_CELL_TARGET=$t M3D_KNN_COOP=1 timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
done
for t in 7; do
  echo "=== M3D_KNN_CELL_TARGET=$t per-lane" >> $L
  M3D_KNN_CELL_TARGET=$t M3D_KNN_COOP=0 timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
done
for t in 14 28; do
  echo "=== bench M3D_KNN_CELL_TARGET=$t coop" >> $L
  M3D_KNN_CELL_TARGET=$t M3D_KNN_COOP=1 timeout 300 python bench.py --steps 40 --warmup 10 --skip-cpu-baseline --skip-roofline --skip-extras 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'])" >> $L 2>&1
done
grep -v amdgpu.ids $L
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
M3D_KNN_CELL_TARGET=20 bash tools/gpu_pmc.sh ${TAG}_t20 "$C" python tools/knn_bench.py pmc | grep -i knn_query
