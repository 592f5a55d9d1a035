#!/bin/bash
# GPU box, round-3 evidence run: full parity suite, the default bench line (all legs, leg-by-leg clock), eager kernel-trace
# stats of the bench command, graph-mode step timeline, HBM-traffic PMC passes of the roofline kernels.
# usage: tools/gpu_r03_final.sh TAG   -> copy gpurun_out/*TAG* into profiles/
set -u
TAG=${1:-r03z}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 600 python -m pytest tests -m gpu -q --timeout 200 --durations=5 2>&1 | tail -14 > $OUT/pytest_gpu_$TAG.log; tail -3 $OUT/pytest_gpu_$TAG.log | cut -c1-200
SECONDS=0
timeout -s KILL 400 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
echo "bench: ${SECONDS}s"; grep -E "^\[bench" $OUT/bench_$TAG.err | tail -3; cut -c1-400 $OUT/bench_$TAG.json; echo
bash tools/gpu_prof.sh $TAG > /dev/null 2>&1
head -8 $OUT/kernel_stats_$TAG.csv | cut -c1-150
bash tools/gpu_trace_analyze.sh $TAG 2>&1 | grep -E "^step:|per queue" | head -3
bash tools/gpu_pmc.sh ${TAG}_fetch "FETCH_SIZE" python tools/pmc_target.py | grep -E "lfa_bwd_kernel<(8|16|64)|knn_query|lfa_fwd_kernel<(8|16)," | cut -c1-160
bash tools/gpu_pmc.sh ${TAG}_write "WRITE_SIZE" python tools/pmc_target.py | grep -E "lfa_bwd_kernel<(8|16|64)|knn_query|lfa_fwd_kernel<(8|16)," | cut -c1-160
