#!/bin/bash
# GPU box: tools/opbench.py WHAT with several variant libraries on one box.  usage: gpu_opbench_libs.sh TAG "old stock d8" wgrad
TAG=$1; LIBS=$2; shift; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in $LIBS; do
  if [ $L = stock ]; then unset M3D_LIB; else export M3D_LIB=$GRAFT_REPO_ROOT/myria3d_amd/variants/libm3d_$L.so; fi
  echo "== $L"; python tools/opbench.py "$@" 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/opbench_$TAG.log
