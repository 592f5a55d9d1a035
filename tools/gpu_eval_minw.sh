#!/bin/bash
# GPU box: same-box A/B of the register caps of the GEMM / weight-gradient kernels (build the variants first:
#   for w in 3 4; do tools/build_variant.sh gemm_rs$w gemm_direct.hip -DGEMM_RS_MINW=$w; tools/build_variant.sh \
#   gemm_kl$w gemm_direct.hip -DGEMM_KL_MINW=$w; tools/build_variant.sh wgrad$w gemm_direct.hip -DWGRAD_MINW=$w; done)
set -u
TAG=${1:-minw}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/eval_$TAG.log; : > $LOG
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
run() { echo "=== $*" >> $LOG; ( eval "$@" ) >> $LOG 2>&1; echo "rc=$?" >> $LOG; }
run "timeout -s KILL 90 python tools/opbench.py gemm wgrad 2>&1 | grep -v Warn"
for lib in $V/libm3d_gemm_*.so $V/libm3d_wgrad*.so; do
  [ -f "$lib" ] || continue
  run "M3D_LIB=$lib timeout -s KILL 90 python tools/opbench.py gemm wgrad 2>&1 | grep -v Warn | tail -1"
  run "M3D_LIB=$lib timeout -s KILL 120 python -m pytest tests/test_gpu_ops.py -q -x -k 'gemm or wgrad or linear or shared' 2>&1 | tail -2"
done
cat $LOG | cut -c1-220
