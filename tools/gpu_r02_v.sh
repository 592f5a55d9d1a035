#!/bin/bash
# fan-out of the per-level position-only queries onto auxiliary streams (M3D_GEO_FANOUT=0: one side stream, as before)
TAG=${1:-r02v}
mkdir -p gpurun_out
L=gpurun_out/geo_fanout_${TAG}.log
: > $L
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -4 >> $L
run() { echo "=== $*" >> $L; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --skip-cpu-baseline --skip-roofline --skip-extras $EXTRA 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'], d['config'].get('geometry_lookahead'))" >> $L 2>&1; }
EXTRA="--lookahead-mode single"; run M3D_GEO_FANOUT=1; run M3D_GEO_FANOUT=0; run M3D_GEO_FANOUT=1; run M3D_GEO_FANOUT=0
EXTRA="--lookahead-mode dual"; run M3D_GEO_FANOUT=1; run M3D_GEO_FANOUT=0
EXTRA="--no-lookahead"; run M3D_GEO_FANOUT=1; run M3D_GEO_FANOUT=0
EXTRA="--no-graph"; run M3D_GEO_FANOUT=1; run M3D_GEO_FANOUT=0
grep -v amdgpu.ids $L
bash tools/gpu_trace_analyze.sh ${TAG} "--lookahead-mode single" > gpurun_out/trace_${TAG}.txt 2>&1; head -4 gpurun_out/trace_${TAG}.txt
