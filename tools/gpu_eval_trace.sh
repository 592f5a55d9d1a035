#!/bin/bash
# GPU box: kernel trace of one graphed eval forward -> gpurun_out/eval_timeline_TAG.csv.   usage: tools/gpu_eval_trace.sh TAG [bf16]
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-e}; MODE=${2:-}
rm -rf /tmp/profe && mkdir -p /tmp/profe
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/profe -o trace -- python $GRAFT_REPO_ROOT/tools/eval_trace.py $MODE ) > $OUT/eval_trace_$TAG.log 2>&1
f=$(find /tmp/profe -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/eval_trace.py --analyze "$f" $OUT/eval_timeline_$TAG.csv
