#!/bin/bash
# GPU box: new preparation kernels + config-5 full-size test, then kNN unroll A/B and the config-5 bench line
set -u
TAG=${1:-prep}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/eval_$TAG.log; : > $LOG
run() { echo "=== $*" >> $LOG; ( eval "$@" ) >> $LOG 2>&1; echo "rc=$?" >> $LOG; }
run "timeout -s KILL 150 python -m pytest tests/test_gpu_prep.py -q -x -s 2>&1 | tail -40"
run "timeout -s KILL 150 python -m pytest tests/test_gpu_net.py -q -x -k dense_tiles 2>&1 | tail -15"
run "timeout -s KILL 200 python -m pytest tests -m gpu -q 2>&1 | tail -8"
run "timeout -s KILL 60 python tools/opbench.py knn 2>&1 | grep -v Warn"
for v in knn_u8 knn_u2; do
  run "M3D_LIB=$GRAFT_REPO_ROOT/myria3d_amd/variants/libm3d_$v.so timeout -s KILL 60 python tools/opbench.py knn 2>&1 | grep -v Warn"
done
run "timeout -s KILL 120 python bench.py --points 40000 --neighbors 32 --skip-roofline --skip-cpu-baseline 2>&1 | tail -1"
cat $LOG | cut -c1-400
