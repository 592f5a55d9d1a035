#!/bin/bash
set -u
TAG=${1:-r02i}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
for i in 1 2; do
M3D_INTERLEAVE_PACED=0 timeout -s KILL 200 python bench.py --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}_a.err | tail -1 | cut -c1-250
M3D_INTERLEAVE_PACED=1 timeout -s KILL 200 python bench.py --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}_b.err | tail -1 | cut -c1-250
done
tail -3 $OUT/bench_${TAG}_b.err
export M3D_INTERLEAVE_PACED=1
bash tools/gpu_trace_analyze.sh ${TAG} 2>&1 | tail -8
awk -F, 'NR>1 && $1<400' $OUT/step_timeline_${TAG}.csv | cut -c1-60 | head -40
