#!/bin/bash
# GPU box: the bf16-activation-storage tests, then the bench's step in the three precision modes (same box, same call).
# usage: tools/gpu_bf16.sh TAG [full]    ("full": also the whole GPU suite)
set -u
TAG=${1:-b}; FULL=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 900 python -m pytest tests/test_gpu_bf16_storage.py -q -s --timeout 400 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" > $OUT/pytest_bf16_$TAG.log
grep -E "passed|failed" $OUT/pytest_bf16_$TAG.log | tail -2 | cut -c1-200
grep -E "^(FAILED|ERROR)|Error|assert |^E  " $OUT/pytest_bf16_$TAG.log | head -60 | cut -c1-300
grep -E "^\[parity\]" $OUT/pytest_bf16_$TAG.log | cut -c1-230
for P in fp32 bf16ops bf16; do
  timeout -s KILL 300 python bench.py --skip-cpu-baseline --skip-roofline --skip-extras --precision $P --steps 100 --warmup 10 2> $OUT/bench_${P}_$TAG.err | tail -1 > $OUT/bench_${P}_$TAG.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${P}_$TAG.json")); print("$P", d["ms_per_step"], "ms/step; eval", d["fwd_only"]["ms_per_step"], "ms; launch", d["config"]["launch"])
except Exception as e:
    print("$P failed", e); print(open("$OUT/bench_${P}_$TAG.err").read()[-1500:])
PY
done
if [ -n "$FULL" ]; then
  timeout -s KILL 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" > $OUT/pytest_gpu_full_$TAG.log
  grep -E "passed|failed" $OUT/pytest_gpu_full_$TAG.log | tail -2 | cut -c1-200
  grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_gpu_full_$TAG.log | head -40 | cut -c1-300
fi
if [ "${3:-}" = "trace" ]; then
  bash tools/gpu_trace_analyze.sh ${TAG}_bf16 "--launch graph --precision bf16" 2>&1 | grep -E "^step:|per queue|^  " | head -16
fi
