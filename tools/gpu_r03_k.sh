#!/bin/bash
set -u
TAG=${1:-r03k}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 400 python -m pytest tests/test_gpu_net.py -m gpu -q --timeout 200 -k "eval_mode_forward_is_differentiable or eval_logits or golden or reference_fixture" 2>&1 | tail -25 | grep -E "passed|failed|FAILED|^E  |parity\] eval" | head -20
