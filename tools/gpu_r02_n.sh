#!/bin/bash
set -u
TAG=${1:-r02n}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "stats\]|passed|failed|Error|assert" | tail -8
timeout -s KILL 200 python tools/opbench.py lfa 2>&1 | grep -i "lfa level" | cut -c1-140
for i in 1 2; do
timeout -s KILL 200 python bench.py --skip-cpu-baseline --skip-extras --skip-roofline 2> $OUT/bench_${TAG}.err | tail -1 | cut -c1-250
done
