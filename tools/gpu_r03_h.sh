#!/bin/bash
# GPU box: kNN grid cell-size sweep (M3D_KNN_CELL_TARGET: points per grid column; any value gives the same exact tables)
set -u
TAG=${1:-r03h}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_cell_$TAG.log; : > $L
for c in 7 3 4 5 6 9 12 16; do
  echo -n "cell_target=$c " | tee -a $L
  M3D_KNN_CELL_TARGET=$c timeout -s KILL 100 python tools/knn_bench.py 2>&1 | tail -1 | sed 's/.*queue=auto: //' | tee -a $L
done
