#!/bin/bash
set -u
TAG=${1:-r03j}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_drain_$TAG.log; : > $L
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 200 -k "knn" 2>&1 | tail -2 | tee -a $L
M3D_LIB=$V/libm3d_dr3.so timeout -s KILL 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 200 -k "knn_self or knn_lidar or k32" 2>&1 | tail -1 | tee -a $L
for rep in 1 2; do
  for lib in default dr1 dr3 dr4 dr4q32; do
    if [ $lib = default ]; then timeout -s KILL 120 python tools/knn_bench.py 2>&1 | tail -1 | tee -a $L
    else M3D_LIB=$V/libm3d_$lib.so timeout -s KILL 120 python tools/knn_bench.py 2>&1 | tail -1 | tee -a $L; fi
  done
done
