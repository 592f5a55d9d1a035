#!/bin/bash
set -u
TAG=${1:-r03i}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
ab() { name=$1; shift; env "$@" timeout -s KILL 100 python bench.py --skip-cpu-baseline --skip-roofline --skip-extras --launch graph 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], 'ms/step')"; }
for rep in 1 2; do
  ab "old dword reduce" M3D_LIB=$GRAFT_REPO_ROOT/myria3d_amd/variants/libm3d_oldreduce.so | tee -a $OUT/ab3_$TAG.log
  for g in 2 4 8 16 32; do ab "float4 reduce gy<=$g" M3D_LFA_RED_GY=$g | tee -a $OUT/ab3_$TAG.log; done
done
