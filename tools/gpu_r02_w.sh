#!/bin/bash
# hardware-queue count (GPU_MAX_HW_QUEUES, ROCclr default 4) x launch mode x geometry fan-out
TAG=${1:-r02w}
mkdir -p gpurun_out
L=gpurun_out/hwq_${TAG}.log
: > $L
run() { echo "=== $* $EXTRA" >> $L; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --skip-cpu-baseline --skip-roofline --skip-extras $EXTRA 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'])" >> $L 2>&1; }
for q in 4 8 16; do
  for f in 0 1; do
    EXTRA="--lookahead-mode single"; run GPU_MAX_HW_QUEUES=$q M3D_GEO_FANOUT=$f
    EXTRA="--lookahead-mode dual"; run GPU_MAX_HW_QUEUES=$q M3D_GEO_FANOUT=$f
  done
  EXTRA="--no-lookahead"; run GPU_MAX_HW_QUEUES=$q M3D_GEO_FANOUT=0
  EXTRA="--no-lookahead"; run GPU_MAX_HW_QUEUES=$q M3D_GEO_FANOUT=1
done
grep -v amdgpu.ids $L
