#!/bin/bash
# GPU box: A/B of the kNN key policies / cell targets / unroll variants on ONE box (parity first, then per-level timing)
# usage: tools/gpu_eval_knn.sh TAG
set -u
TAG=${1:-knn}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/knn_eval_$TAG.log; : > $LOG
run() { echo "=== $*" >> $LOG; ( eval "$@" ) >> $LOG 2>&1; echo "rc=$?" >> $LOG; }
run "timeout -s KILL 200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4"
run "M3D_KNN_KEYS=f64 timeout -s KILL 200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4"
for cfg in "u64 7" "f64 7" "f64 10" "f64 5" "f64 14"; do
  set -- $cfg
  run "M3D_KNN_KEYS=$1 M3D_KNN_CELL_TARGET=$2 timeout -s KILL 60 python tools/opbench.py knn 2>&1 | grep -v Warn"
done
for v in knn_u8 knn_u2; do
  run "M3D_LIB=$GRAFT_REPO_ROOT/myria3d_amd/variants/libm3d_$v.so M3D_KNN_KEYS=f64 timeout -s KILL 60 python tools/opbench.py knn 2>&1 | grep -v Warn"
done
cat $LOG
