#!/bin/bash
# GPU box: debug call — dual-graph NaN localisation; forced-collective bench by launch mode.  usage: tools/gpu_r03_d.sh TAG
set -u
TAG=${1:-r03d}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 200 python tools/scratch/dual_debug.py 2>&1 | grep -vE "amdgpu.ids" | tail -45 | tee $OUT/dual_debug_$TAG.log
for L in eager graph; do
  SECONDS=0
  timeout -s KILL 90 python bench.py --force-collective --launch $L --skip-cpu-baseline --skip-roofline --skip-extras --steps 5 --warmup 2 > $OUT/coll_${L}_$TAG.json 2> $OUT/coll_${L}_$TAG.err
  echo "collective launch=$L rc=$? ${SECONDS}s: $(grep -vE 'amdgpu.ids|socket.cpp' $OUT/coll_${L}_$TAG.err | tail -2 | cut -c1-200)"; tail -c 300 $OUT/coll_${L}_$TAG.json; echo
done
