#!/bin/bash
# GPU box, round-4 evidence run: full parity suite, the default bench line (all legs), eager kernel-trace stats of the bench
# command, graph-mode step timeline, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE) and an SQ instruction-count pass of the
# roofline kernels.  usage: tools/gpu_r04_final.sh TAG   -> copy gpurun_out/*TAG* into profiles/
set -u
TAG=${1:-r04fin}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 400 --durations=5 2>&1 | grep -v "^  File\|^Extension modules" | tail -14 > $OUT/pytest_gpu_$TAG.log; tail -3 $OUT/pytest_gpu_$TAG.log | cut -c1-200
SECONDS=0
timeout -s KILL 600 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
echo "bench: ${SECONDS}s"; grep -E "^\[bench" $OUT/bench_$TAG.err > $OUT/bench_progress_$TAG.log; tail -2 $OUT/bench_progress_$TAG.log; cut -c1-300 $OUT/bench_$TAG.json; echo
bash tools/gpu_prof.sh $TAG > /dev/null 2>&1
head -8 $OUT/kernel_stats_$TAG.csv | cut -c1-150
bash tools/gpu_trace_analyze.sh $TAG 2>&1 | grep -E "^step:|per queue" | head -3
bash tools/gpu_pmc.sh ${TAG}_fetch "FETCH_SIZE" python tools/pmc_target.py | grep -E "lfa_bwd_kernel<(8|16|64)|knn_query|lfa_fwd_kernel<(8|16)," | cut -c1-160
bash tools/gpu_pmc.sh ${TAG}_write "WRITE_SIZE" python tools/pmc_target.py | grep -E "lfa_bwd_kernel<(8|16|64)|knn_query|lfa_fwd_kernel<(8|16)," | cut -c1-160
bash tools/gpu_pmc.sh ${TAG}_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" python tools/pmc_target.py | grep -E "lfa_bwd_kernel<(8|16|64)|knn_query|lfa_fwd_kernel<(8|16)," | cut -c1-260
