#!/bin/bash
# GPU box: same-box A/B of an import-time switch (environment variable) on the bench's training step.
# usage: tools/gpu_env_ab.sh TAG VAR "v1 v2 ..." "fp32 bf16" [repeats]
set -u
TAG=${1:-ab}; VAR=$2; VALS=$3; PRECS=${4:-"fp32"}; REP=${5:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/envab_$TAG.log
for r in $(seq 1 $REP); do
 for V in $VALS; do
  for P in $PRECS; do
    export $VAR=$V
    timeout -s KILL 300 python bench.py --skip-cpu-baseline --skip-roofline --skip-extras --precision $P --steps 200 --warmup 10 --launch graph 2> $OUT/ab_err.log | tail -1 > $OUT/ab_line.json
    python - <<PY | tee -a $OUT/envab_$TAG.log
import json
try:
    d=json.load(open("$OUT/ab_line.json")); print("run $r $VAR=$V prec $P:", d["ms_per_step"], "ms/step; eval", d["fwd_only"]["ms_per_step"])
except Exception as e:
    print("run $r $VAR=$V prec $P failed", e); print(open("$OUT/ab_err.log").read()[-800:])
PY
  done
 done
done
