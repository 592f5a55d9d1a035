#!/bin/bash
# GPU box: second A/B of the LFA backward: level-1 tile / occupancy variants against the new default, same box
set -u
TAG=${1:-bwd2}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/eval_$TAG.log; : > $LOG
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
run() { echo "=== $*" >> $LOG; ( eval "$@" ) >> $LOG 2>&1; echo "rc=$?" >> $LOG; }
T='python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_train.py -q -x -k "lfa or net or train or oracle or golden or full" 2>&1 | tail -3'
run "timeout -s KILL 150 $T"
run "timeout -s KILL 60 python tools/opbench.py lfa 2>&1 | grep '^lfa level [12]'"
for n in p16w2 w16_2 r128w4 r128w4p r128w3p; do
  run "M3D_LIB=$V/libm3d_bwd_$n.so timeout -s KILL 60 python tools/opbench.py lfa 2>&1 | grep '^lfa level 1'"
done
run "M3D_LIB=$V/libm3d_bwd_r128w4p.so timeout -s KILL 150 $T"
run "M3D_LIB=$V/libm3d_bwd_r128w4.so timeout -s KILL 150 $T"
cat $LOG | cut -c1-200
