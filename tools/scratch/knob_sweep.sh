#!/bin/bash
# GPU box: launch-geometry knobs inside the step (in-step timeline = dependent chain, one box): main-queue microseconds
run() {  # name, env assignments...
  name=$1; shift
  env "$@" bash tools/gpu_trace_analyze.sh ks_$name > /dev/null 2>&1
  python - "$name" <<'PY'
import csv, sys, os, collections
rows = [r for r in csv.DictReader(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"step_timeline_ks_{sys.argv[1]}.csv"))) if r["queue"] == "1"]
agg = collections.defaultdict(float)
for r in rows:
    k = r["kernel"]
    fam = "gemm" if "gemm_" in k else ("bn" if k.startswith("bn_") else ("wgrad" if "wgrad" in k else ("lfa" if "lfa_" in k else "other")))
    agg[fam] += float(r["dur_us"])
print(f"{sys.argv[1]:28s} main {sum(agg.values()):7.1f} us  " + "  ".join(f"{k} {v:6.1f}" for k, v in sorted(agg.items())))
PY
}
run default X=1
run rs_cap_1024 M3D_GEMM_RS_CAP=1024
run rs_cap_512 M3D_GEMM_RS_CAP=512
run kl_minwaves_1024 M3D_GEMM_KL_MINWAVES=1024
run kl_minwaves_2048 M3D_GEMM_KL_MINWAVES=2048
run bn_slots_0_4 M3D_BN_SLOTS=0:4
run bn_slots_16_4 M3D_BN_SLOTS=100000:16,0:4
run bwd_slots_8_2 M3D_BN_BWD_SLOTS=100000:8,0:2
run bwd_slots_4_4 M3D_BN_BWD_SLOTS=0:4
run bwd_slots_16_4 M3D_BN_BWD_SLOTS=100000:16,0:4
run default_again X=1
