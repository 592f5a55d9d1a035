#!/bin/bash
# GPU box: LFA backward workgroup geometry for the narrow layers (ch 8 / 16 / 32): waves per workgroup x edge rows per trip
set -u
TAG=${1:-r03l}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/lfa_bwd_geom_$TAG.log; : > $L
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
echo "== default" | tee -a $L; timeout -s KILL 100 python tools/opbench.py lfa 2>&1 | grep '^lfa level [12]' | tee -a $L
for lib in g1 g2 g3 g4 g6; do
  echo "== $lib" | tee -a $L
  M3D_LIB=$V/libm3d_$lib.so timeout -s KILL 100 python tools/opbench.py lfa 2>&1 | grep '^lfa level [12]' | tee -a $L
  M3D_LIB=$V/libm3d_$lib.so timeout -s KILL 150 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 120 -k "lfa_backward or lfa_train or lfa_bf16" 2>&1 | tail -1 | tee -a $L
done
