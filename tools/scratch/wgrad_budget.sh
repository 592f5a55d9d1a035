#!/bin/bash
# GPU box: wave budgets of the batched weight-gradient launch inside the step (in-step timeline, one box)
for cfg in "4096 8192" "6144 8192" "8192 8192" "4096 12288" "3072 6144"; do
  set -- $cfg
  M3D_WGRAD_BATCH_WAVES_BIG=$1 M3D_WGRAD_BATCH_WAVES_SMALL=$2 bash tools/gpu_trace_analyze.sh wg_$1_$2 > /dev/null 2>&1
  python - "$1" "$2" <<'PY'
import csv, sys, os
rows = [r for r in csv.DictReader(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"step_timeline_wg_{sys.argv[1]}_{sys.argv[2]}.csv"))) if r["queue"] == "1"]
w = [round(float(r["dur_us"]), 1) for r in rows if "wgrad" in r["kernel"]]
print(f"big {sys.argv[1]} small {sys.argv[2]}: main {sum(float(r['dur_us']) for r in rows):.1f} us, wgrad kernels {sum(w):.1f} us {w}")
PY
done
