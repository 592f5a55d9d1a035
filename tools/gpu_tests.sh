#!/bin/bash
# GPU box: the GPU parity suite with its margins (no bench).  usage: tools/gpu_tests.sh TAG [pytest args]
set -u
TAG=${1:-t}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 1500 python -m pytest tests -m gpu -q -s --timeout 600 --durations=8 "$@" 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" > $OUT/pytest_gpu_full_$TAG.log
grep -E "passed|failed" $OUT/pytest_gpu_full_$TAG.log | tail -2 | cut -c1-200
grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest_gpu_full_$TAG.log | head -40 | cut -c1-300
grep -E "^\[parity\]" $OUT/pytest_gpu_full_$TAG.log | cut -c1-230 > $OUT/parity_margins_$TAG.log
grep -E "GraphedStep|above 0.001|worst \(HIP" -A7 $OUT/parity_margins_$TAG.log | head -150
