#!/bin/bash
# GPU box: software-pipelined kNN ring walk — parity (the default build), then same-box A/B per level against the
# run-by-run variants (tools/build_variant.sh knn_nopipe knn.hip -DKNNQ_PIPE=0 -DKNN_PIPE=0; knn_qpipe: queue kernel only)
set -u
TAG=${1:-r03g}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_pipe_$TAG.log; : > $L
timeout -s KILL 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --timeout 150 -k "knn or batched or full_size or interpolate or golden or eval_logits" 2>&1 | tail -4 | tee -a $L
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
for rep in 1 2; do
  for lib in knn_nopipe knn_qpipe default; do
    if [ $lib = default ]; then timeout -s KILL 120 python tools/knn_bench.py 2>&1 | tail -1 | tee -a $L
    else M3D_LIB=$V/libm3d_$lib.so timeout -s KILL 120 python tools/knn_bench.py 2>&1 | tail -1 | tee -a $L; fi
  done
done
