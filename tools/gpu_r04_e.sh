#!/bin/bash
# GPU box, round 4 call E: full suite, FPS timing (bucket v2), the whole bench line with the new legs (end-to-end predict,
# single-threaded autograd, CPU baseline on 16 tiles), eager-launch probe.
set -u
TAG=${1:-r04e}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 400 --durations=6 2>&1 | grep -v "^  File\|^Extension modules" > $OUT/pytest_gpu_full_$TAG.log
(head -20 $OUT/pytest_gpu_full_$TAG.log; echo ...; tail -16 $OUT/pytest_gpu_full_$TAG.log) | cut -c1-240 > $OUT/pytest_gpu_$TAG.log; tail -10 $OUT/pytest_gpu_$TAG.log
timeout -s KILL 200 python bench.py --mode pointnet2 --steps 3 --tiles 16 --points 40000 --neighbors 32 2>/dev/null | tail -1 > $OUT/pointnet2_$TAG.json; cut -c1-360 $OUT/pointnet2_$TAG.json; echo
SECONDS=0
timeout -s KILL 600 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
echo "bench: ${SECONDS}s"; grep -E "^\[bench" $OUT/bench_$TAG.err | tail -20; python - <<PY
import json
d=json.load(open("$OUT/bench_$TAG.json"))
for k in ("ms_per_step","eager_ms_per_step","dropin_eager_ms_per_step","dropin_variable_layout_ms_per_step","optin_variable_layout_ms_per_step","optin_variable_layout_single_thread_autograd_ms_per_step","dropin_variable_layout_single_thread_autograd_ms_per_step"):
    print(k, d.get(k))
print("config", d["config"])
print("fwd_only", d.get("fwd_only"))
print("bf16", (d.get("bf16") or {}).get("ms_per_step"))
print("collective", d.get("forced_collective_1rank"))
print("pointnet2", {k: v for k, v in (d.get("pointnet2_config5") or {}).items() if k.endswith("ms") or k == "ms_per_step"})
print("predict", d.get("predict_config3"))
print("predict e2e", d.get("predict_config3_end_to_end"))
print("cpu", d.get("cpu_baseline"))
print("launch_probe", d.get("launch_probe_ms"))
PY
