#!/bin/bash
# GPU box: same-box A/B of variant libraries (myria3d_amd/variants/libm3d_NAME.so; "stock" = the product library) on the
# bench's training step.  usage: tools/gpu_ab.sh TAG "stock ut1 ..." "fp32 bf16" [repeats]
set -u
TAG=${1:-ab}; LIBS=${2:-stock}; PRECS=${3:-"fp32 bf16"}; REP=${4:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/ab_$TAG.log
for r in $(seq 1 $REP); do
 for L in $LIBS; do
  for P in $PRECS; do
    if [ "$L" = stock ]; then unset M3D_LIB; else export M3D_LIB=$GRAFT_REPO_ROOT/myria3d_amd/variants/libm3d_$L.so; fi
    timeout -s KILL 300 python bench.py --skip-cpu-baseline --skip-roofline --skip-extras --precision $P --steps 200 --warmup 10 --launch graph 2> $OUT/ab_err.log | tail -1 > $OUT/ab_line.json
    python - <<PY | tee -a $OUT/ab_$TAG.log
import json
try:
    d=json.load(open("$OUT/ab_line.json")); print("run $r lib $L prec $P:", d["ms_per_step"], "ms/step; eval", d["fwd_only"]["ms_per_step"])
except Exception as e:
    print("run $r lib $L prec $P failed", e); print(open("$OUT/ab_err.log").read()[-800:])
PY
  done
 done
done
