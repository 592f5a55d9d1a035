#!/bin/bash
# GPU box, round-2 evidence run: full parity suite, the default bench line (all legs), eager kernel-trace stats,
# HBM-traffic PMC passes of the roofline kernels.  usage: tools/gpu_r02_final.sh TAG
set -u
TAG=${1:-r02z}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -14 > $OUT/pytest_gpu_$TAG.log; tail -4 $OUT/pytest_gpu_$TAG.log
timeout -s KILL 600 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
cut -c1-1500 $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
bash tools/gpu_prof.sh $TAG > /dev/null 2>&1
head -12 $OUT/kernel_stats_$TAG.csv | cut -c1-160
bash tools/gpu_pmc.sh ${TAG}_fetch "FETCH_SIZE" python tools/pmc_target.py | grep -E "lfa_bwd_kernel<64|knn_query_queue|lfa_fwd_kernel<16" | cut -c1-200
bash tools/gpu_pmc.sh ${TAG}_write "WRITE_SIZE" python tools/pmc_target.py | grep -E "lfa_bwd_kernel<64|knn_query_queue|lfa_fwd_kernel<16" | cut -c1-200
