#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, the bench line, and a rocprofv3 kernel-trace of the bench.
# Outputs land under gpurun_out/ (merged back); summaries worth judging are copied to profiles/ by hand.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
TAG=${1:-r01}
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest_gpu_$TAG.log 2>&1
( timeout 600 python bench.py 2> $OUT/bench_$TAG.err | tail -3 ) > $OUT/bench_$TAG.json
( timeout 600 python bench.py --no-graph --skip-cpu-baseline 2>> $OUT/bench_$TAG.err | tail -3 ) > $OUT/bench_nograph_$TAG.json
rm -rf /tmp/prof && mkdir -p /tmp/prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 5 --warmup 2 --skip-cpu-baseline ) > $OUT/rocprof_$TAG.log 2>&1
find /tmp/prof -name '*stats*' -o -name '*kernel_stats*' | head
for f in $(find /tmp/prof -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats_$TAG.csv; done
for f in $(find /tmp/prof -name '*domain_stats.csv'); do cp $f $OUT/domain_stats_$TAG.csv; done
cat $OUT/pytest_gpu_$TAG.log | tail -5
cat $OUT/bench_$TAG.json
cat $OUT/bench_nograph_$TAG.json
head -40 $OUT/kernel_stats_$TAG.csv
