#!/bin/bash
# GPU box: third A/B of the LFA backward: finer workgroups + occupancy caps for ch = 32 / 64 / 128, same box
set -u
TAG=${1:-bwd3}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/eval_$TAG.log; : > $LOG
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
run() { echo "=== $*" >> $LOG; ( eval "$@" ) >> $LOG 2>&1; echo "rc=$?" >> $LOG; }
T='python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_train.py -q -x -k "lfa or net or train or oracle or golden or full" 2>&1 | tail -3'
run "timeout -s KILL 150 $T"
run "timeout -s KILL 60 python tools/opbench.py lfa 2>&1 | grep '^lfa'"
for n in v32 v64 v128 v128b; do
  run "M3D_LIB=$V/libm3d_bwd_$n.so timeout -s KILL 60 python tools/opbench.py lfa 2>&1 | grep '^lfa level [234]'"
  run "M3D_LIB=$V/libm3d_bwd_$n.so timeout -s KILL 150 $T"
done
cat $LOG | cut -c1-200
