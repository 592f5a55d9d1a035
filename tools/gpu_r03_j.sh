#!/bin/bash
# GPU box: ping-pong candidate loops (real prefetch) — parity with the default build, then per-level A/B
set -u
TAG=${1:-r03j}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_pp_$TAG.log; : > $L
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --timeout 200 -k "knn or batched or full_size or interpolate or golden or eval_logits" 2>&1 | tail -2 | tee -a $L
for rep in 1 2; do
  for lib in pp0 qpp default pp_u2; do
    if [ $lib = default ]; then timeout -s KILL 120 python tools/knn_bench.py 2>&1 | tail -1 | tee -a $L
    else M3D_LIB=$V/libm3d_$lib.so timeout -s KILL 120 python tools/knn_bench.py 2>&1 | tail -1 | tee -a $L; fi
  done
done
