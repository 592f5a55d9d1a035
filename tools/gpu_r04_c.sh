#!/bin/bash
# GPU box, round 4 call C: the whole parity suite in ONE process (as the driver runs it), the N > 1 launch-form probe after the
# stream fix, the default bench line, host profile of the variable-layout step.
set -u
TAG=${1:-r04c}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 400 --durations=8 2>&1 | grep -v "^  File\|^Extension modules" > $OUT/pytest_gpu_full_$TAG.log
(head -30 $OUT/pytest_gpu_full_$TAG.log; echo ...; tail -25 $OUT/pytest_gpu_full_$TAG.log) | cut -c1-240 > $OUT/pytest_gpu_$TAG.log; tail -14 $OUT/pytest_gpu_$TAG.log
for c in none pg captured eager; do timeout -s KILL 150 python tools/collective_probe.py $c 2>&1 | grep collective_probe; done > $OUT/collective_probe_$TAG.log; cat $OUT/collective_probe_$TAG.log
timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench | tee $OUT/knn_$TAG.log
SECONDS=0
timeout -s KILL 420 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
echo "bench: ${SECONDS}s"; python - <<PY
import json
d=json.load(open("$OUT/bench_$TAG.json"))
for k in ("ms_per_step","eager_ms_per_step","dropin_eager_ms_per_step","dropin_variable_layout_ms_per_step","optin_variable_layout_ms_per_step"):
    print(k, d.get(k))
print("config", d["config"])
print("fwd_only", d.get("fwd_only",{}).get("ms_per_step"))
print("bf16", (d.get("bf16") or {}).get("ms_per_step"))
print("collective", d.get("forced_collective_1rank"))
print("pointnet2", {k: v for k, v in (d.get("pointnet2_config5") or {}).items() if k.endswith("ms") or k == "ms_per_step"})
print("predict", d.get("predict_config3"))
for e in d.get("roofline_knn_lse_stage", []): print("  ", e["kernel"][:60], e["avg_launch_ms"], e["frac"])
PY
timeout -s KILL 200 python tools/host_profile.py variable 2>&1 | head -60 > $OUT/host_profile_variable_$TAG.log; head -50 $OUT/host_profile_variable_$TAG.log | cut -c1-160
