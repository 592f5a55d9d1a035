#!/bin/bash
set -u
TAG=${1:-r02e}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 600 python -m pytest tests -m gpu -q -x -k "lookahead or seeded or tile_select or forward_like or fused_adam or poisons or gradients_are_ready or persistent" --durations=8 2>&1 | tail -25
for i in 1 2; do
timeout -s KILL 200 python bench.py --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}_base.err | tail -1 | cut -c1-420
timeout -s KILL 200 python bench.py --lookahead --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}_look.err | tail -1 | cut -c1-420
done
tail -5 $OUT/bench_${TAG}_look.err
