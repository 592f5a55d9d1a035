#!/bin/bash
# GPU box: the bench's two predict legs (in the bench process, behind the contract line's timing) with and without the
# one-batch lookahead of predict_cloud.  usage: gpu_predict_bench_ab.sh TAG [repeats]
TAG=${1:-pab}; REP=${2:-2}
cd $GRAFT_REPO_ROOT; OUT=gpurun_out; mkdir -p $OUT; : > $OUT/predict_bench_ab_$TAG.log
for r in $(seq 1 $REP); do for L in 0 1; do
  M3D_PREDICT_LOOKAHEAD=$L timeout 600 python bench.py --skip-cpu-baseline --skip-roofline --skip-legs bf16,bf16x3,dropin,collective,torch,dense,pointnet2 2>/dev/null | tail -1 > $OUT/pab_line.json
  python - <<PY | tee -a $OUT/predict_bench_ab_$TAG.log
import json
d=json.load(open("$OUT/pab_line.json")); print("run $r lookahead=$L: step", d["ms_per_step"], "predict sweep", d["predict_config3"]["ms_per_sweep"], "ms; chain", d["predict_config3_end_to_end"]["ms_per_cloud"], "ms")
PY
done; done
python tools/predict_ab.py 3 50 2>&1 | grep "batch_size" | tee -a $OUT/predict_bench_ab_$TAG.log
