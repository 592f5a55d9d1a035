#!/bin/bash
# GPU box: SQ counters (one --pmc pass) of the roofline kernels: how much of a wavefront's life is instruction issue
set -u
TAG=${1:-r03m}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
bash tools/gpu_pmc.sh ${TAG}_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" python tools/pmc_target.py | grep -E "lfa_bwd_kernel<(8|16|64)|knn_query|lfa_fwd_kernel<(8|16)," | cut -c1-260
bash tools/gpu_pmc.sh ${TAG}_sq2 "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" python tools/pmc_target.py | grep -E "lfa_bwd_kernel<(8|16|64)|knn_query_queue|lfa_fwd_kernel<(8|16)," | cut -c1-260
