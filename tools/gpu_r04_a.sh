#!/bin/bash
# GPU box, round 4 call A: full parity suite, kNN kernel A/B (trim / network / register cap), level-1 LFA-backward phase ablation,
# the default bench line.  usage: tools/gpu_r04_a.sh TAG
set -u
TAG=${1:-r04a}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 700 python -m pytest tests -m gpu -q --timeout 300 --durations=8 -x 2>&1 | tail -40 > $OUT/pytest_gpu_$TAG.log; tail -4 $OUT/pytest_gpu_$TAG.log | cut -c1-250
{
  timeout -s KILL 120 python tools/knn_bench.py
  for v in knn_notrim knn_nonet knn_r3like knn_minw4 knn_u2; do M3D_LIB=$V/libm3d_$v.so timeout -s KILL 120 python tools/knn_bench.py; done
} 2>&1 | grep -E "knn_bench|Error|error" > $OUT/knn_ab_$TAG.log; cat $OUT/knn_ab_$TAG.log
{
  echo "== default"; timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level (1|2)"
  for v in bwd_nopipe bwd_dbg1 bwd_dbg2 bwd_dbg4 bwd_dbg8 bwd_dbg16 bwd_dbg32; do echo "== $v"; M3D_LIB=$V/libm3d_$v.so timeout -s KILL 200 python tools/opbench.py lfa | grep -E "lfa level 1"; done
} > $OUT/lfa_bwd_l1_ablation_$TAG.log 2>&1; cat $OUT/lfa_bwd_l1_ablation_$TAG.log
SECONDS=0
timeout -s KILL 420 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
echo "bench: ${SECONDS}s"; grep -E "^\[bench" $OUT/bench_$TAG.err | tail -3; cut -c1-1500 $OUT/bench_$TAG.json; echo
python - <<PY
import json
d=json.load(open("$OUT/bench_$TAG.json"))
for k in ("ms_per_step","eager_ms_per_step","dropin_eager_ms_per_step","dropin_variable_layout_ms_per_step","optin_variable_layout_ms_per_step"):
    print(k, d.get(k))
print("fwd_only", d.get("fwd_only",{}).get("ms_per_step"))
print("bf16", d.get("bf16"))
print("collective", d.get("forced_collective_1rank"))
print("pointnet2", d.get("pointnet2_config5"))
print("roofline", d.get("roofline"))
for e in d.get("roofline_knn_lse_stage", []): print("  ", e)
PY
