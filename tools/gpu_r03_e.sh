#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
for V in "--points 12800" "--points 3200"; do
  SECONDS=0
  timeout -s KILL 75 python bench.py --force-collective --launch graph --lookahead-mode dual $V --skip-cpu-baseline --skip-roofline --skip-extras --steps 5 --warmup 2 > $OUT/coll_v.json 2> $OUT/coll_v.err
  echo "collective graph dual [$V] rc=$? ${SECONDS}s: $(grep -vE 'amdgpu.ids|socket.cpp|RCCL|HIP version|ROCm version|Hostname|Librccl' $OUT/coll_v.err | tail -2 | cut -c1-160 | tr '\n' ' ')"; tail -c 200 $OUT/coll_v.json; echo
done
