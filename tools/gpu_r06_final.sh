#!/bin/bash
# GPU box, round-6 evidence run (ONE per round): full parity suite (margins printed), the default bench line (all legs), eager
# kernel-trace stats of the bench command, graph-mode step timeline, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE) and two SQ
# passes over the roofline kernels.   usage: tools/gpu_r06_final.sh TAG   -> copy gpurun_out/*TAG* into profiles/
set -u
TAG=${1:-r06fin}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 1100 python -m pytest tests -m gpu -q -s --timeout 400 --durations=8 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" > $OUT/pytest_gpu_full_$TAG.log
grep -E "passed|failed" $OUT/pytest_gpu_full_$TAG.log | tail -2 | cut -c1-200
{ grep -E "passed|failed|slowest|^[0-9.]+s (call|setup)" $OUT/pytest_gpu_full_$TAG.log | tail -12; } > $OUT/pytest_gpu_$TAG.log
grep -E "^\[parity\]" $OUT/pytest_gpu_full_$TAG.log | cut -c1-230 > $OUT/parity_margins_$TAG.log
SECONDS=0
timeout -s KILL 900 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
echo "bench: ${SECONDS}s"; grep -E "^\[bench" $OUT/bench_$TAG.err > $OUT/bench_progress_$TAG.log; tail -2 $OUT/bench_progress_$TAG.log; cut -c1-400 $OUT/bench_$TAG.json; echo
bash tools/gpu_prof.sh $TAG > /dev/null 2>&1
head -8 $OUT/kernel_stats_$TAG.csv | cut -c1-150
bash tools/gpu_trace_analyze.sh $TAG "--launch graph" 2>&1 | grep -E "^step:|per queue" | head -3
K='lfa_bwd_kernel<64|lfa_bwd_small_kernel|gather_sum_rows4|knn_query|lfa_fwd_full_kernel<(8|16),'
bash tools/gpu_pmc.sh ${TAG}_fetch "FETCH_SIZE" python tools/pmc_target.py | grep -E "$K" | cut -c1-160
bash tools/gpu_pmc.sh ${TAG}_write "WRITE_SIZE" python tools/pmc_target.py | grep -E "$K" | cut -c1-160
bash tools/gpu_pmc.sh ${TAG}_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" python tools/pmc_target.py | grep -E "$K" | cut -c1-260
bash tools/gpu_pmc.sh ${TAG}_sq2 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT" python tools/pmc_target.py | grep -E "$K" | cut -c1-260
# round 6: the "bf16" leg (bf16 activation storage + bf16 matrix-core operands) kernel by kernel
bash tools/gpu_trace_analyze.sh ${TAG}_bf16 "--launch graph --precision bf16" 2>&1 | grep -E "^step:|per queue" | head -3
