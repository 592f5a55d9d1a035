#!/bin/bash
# GPU box: staged kNN — parity, then per-level timing (hipGraph replay, tools/knn_bench.py) of the single-launch kernels
# against stage schedules.  usage: tools/gpu_r03_knn.sh TAG
set -u
TAG=${1:-r03k}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_staged_$TAG.log; : > $L
timeout -s KILL 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "knn or batched" 2>&1 | tail -8 | tee -a $L
run() { echo "--- $*" >> $L; env "$@" timeout -s KILL 120 python tools/knn_bench.py 2>&1 | tail -1 | tee -a $L; }
run M3D_KNN_STAGED=0
run M3D_KNN_STAGED=0 M3D_KNN_QUEUE=1
run M3D_X=default
for S in "2,4:3,8:4,16" "1,2:2,4:3,16" "2,16" "2,4:3,16" "1,4:2,8:3,16" "2,2:3,4:4,16" "3,8:4,16" "2,8:3,16:4,16"; do
  run M3D_KNN_STAGED=1 M3D_KNN_STAGES="$S"
done
run M3D_KNN_STAGED=1 M3D_KNN_STAGE_GRID=1024
run M3D_KNN_STAGED=1 M3D_KNN_STAGE_GRID=16384
cat $L | grep -E "^knn_bench|^---|passed|failed" | tail -40
