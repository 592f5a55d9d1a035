#!/bin/bash
set -u
TAG=${1:-r02l}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 600 python -m pytest tests -m gpu -q -x -k "bf16 or gemm or shared_layer or residual or train_forward or eval_logits" 2>&1 | grep -E "parity\]|passed|failed|Error|assert" | tail -14
timeout -s KILL 400 python bench.py --skip-cpu-baseline --skip-roofline 2> $OUT/bench_${TAG}.err | tail -1 > $OUT/bench_${TAG}.json
python - <<PY
import json
d=json.load(open("$OUT/bench_${TAG}.json"))
print("fp32", d["ms_per_step"], d["fwd_only"]["ms_per_step"], "eager", d.get("eager_ms_per_step"))
print("bf16", d.get("bf16"))
PY
