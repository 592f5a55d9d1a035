#!/bin/bash
# GPU box: counter passes (rocprofv3 --pmc only) over the eager training step, weight-gradient kernels only.
# usage: gpu_pmc_wgrad.sh TAG
TAG=${1:-r06zd}
cd $GRAFT_REPO_ROOT
CMD="python bench.py --no-graph --steps 3 --warmup 1 --skip-cpu-baseline --skip-extras --skip-roofline"
bash tools/gpu_pmc.sh ${TAG}_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" $CMD | grep -E "wgrad" | cut -c1-400
bash tools/gpu_pmc.sh ${TAG}_b "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum" $CMD | grep -E "wgrad" | cut -c1-400
bash tools/gpu_pmc.sh ${TAG}_c "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" $CMD | grep -E "wgrad" | cut -c1-400
