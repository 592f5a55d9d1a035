#!/bin/bash
# copies the evidence files of a tools/gpu_r06_final.sh run from gpurun_out/ into profiles/ under the names bench.py and
# profiles/README.md use.  usage: tools/collect_profiles.sh TAG
set -e
T=$1; G=gpurun_out; P=profiles
cp $G/bench_$T.json $P/${T}_bench.json; cp $G/bench_progress_$T.log $P/${T}_bench_progress.log
cp $G/kernel_stats_$T.csv $P/${T}_kernel_stats.csv; cp $G/kernel_trace_${T}_summary.csv $P/${T}_kernel_trace_summary.csv
cp $G/parity_margins_$T.log $P/${T}_parity_margins.log; cp $G/pytest_gpu_$T.log $P/${T}_pytest_gpu.log
cp $G/pmc_${T}_fetch.csv $P/${T}_pmc_fetch_size.csv; cp $G/pmc_${T}_write.csv $P/${T}_pmc_write_size.csv
cp $G/pmc_${T}_sq.csv $P/${T}_pmc_sq_roofline_kernels.csv; cp $G/pmc_${T}_sq2.csv $P/${T}_pmc_sq2_roofline_kernels.csv
cp $G/step_timeline_$T.csv $P/${T}_step_timeline.csv
[ -f $G/step_timeline_${T}_bf16.csv ] && cp $G/step_timeline_${T}_bf16.csv $P/${T}_step_timeline_bf16.csv
grep -E "^step:|per queue|^  |main queue|queue [0-9]" $G/trace_$T.log > $P/${T}_trace.log || true
ls $P | grep "^$T" | wc -l
