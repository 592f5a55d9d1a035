#!/bin/bash
# weight gradients deferred to the end of the backward pass and launched per tile class (M3D_DEFER_WGRAD=0: per layer)
TAG=${1:-r02y}
mkdir -p gpurun_out
L=gpurun_out/wgrad_batch_${TAG}.log
: > $L
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 >> $L
run() { echo "=== $* $EXTRA" >> $L; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --skip-cpu-baseline --skip-roofline --skip-extras $EXTRA 2>>$L | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'], d['config']['launch'], d.get('launch_probe_ms'))" >> $L 2>&1; }
EXTRA="--launch graph"; run M3D_DEFER_WGRAD=1; run M3D_DEFER_WGRAD=0; run M3D_DEFER_WGRAD=1; run M3D_DEFER_WGRAD=0
EXTRA="--launch eager --warmup 45"; run M3D_DEFER_WGRAD=1; run M3D_DEFER_WGRAD=0
EXTRA="--launch graph --no-lookahead"; run M3D_DEFER_WGRAD=1; run M3D_DEFER_WGRAD=0
grep -v amdgpu.ids $L
bash tools/gpu_trace_analyze.sh ${TAG} "--launch graph" > gpurun_out/trace_${TAG}.txt 2>&1; head -16 gpurun_out/trace_${TAG}.txt
