#!/bin/bash
# GPU box: kNN parity (both query kernels) + same-box A/B timing.  usage: tools/gpu_r02_knn.sh TAG
set -u
TAG=${1:-r02b}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_ab_$TAG.log; : > $L
for q in 1 0; do
  echo "=== M3D_KNN_QUEUE=$q: parity" >> $L
  M3D_KNN_QUEUE=$q timeout -s KILL 600 python -m pytest tests -m gpu -q -x -k "knn or eval_logits or full_size or seeded or golden or interpolate" 2>&1 | tail -5 >> $L
  echo "=== M3D_KNN_QUEUE=$q: opbench knn" >> $L
  M3D_KNN_QUEUE=$q timeout -s KILL 200 python tools/opbench.py knn 2>&1 | grep -i "knn" >> $L
done
cat $L
