#!/bin/bash
# GPU box: accumulate-in-accumulator dgrad (GemmArgs::acc_pre): parity, per-shape timings, the training step
set -u
TAG=${1:-r03n}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_train.py -m gpu -x -q \
  -k "gemm or dgrad or linear or shared_input or flattened or graphed_step or bf16 or fp_module or train_steps" 2>&1 | tail -5
timeout -s KILL 200 python tools/opbench.py gemm bn > $OUT/opbench_${TAG}.log 2>&1; cat $OUT/opbench_${TAG}.log
M3D_GEMM_RS_CAP=1536 timeout -s KILL 200 python tools/opbench.py gemm > $OUT/opbench_${TAG}_cap1536.log 2>&1; grep -E "M=204800|M= 51200|TOTAL" $OUT/opbench_${TAG}_cap1536.log
timeout -s KILL 200 python bench.py --steps 30 --warmup 8 --skip-cpu-baseline --skip-roofline --skip-extras 2>/dev/null | tail -1 | cut -c1-400
