#!/bin/bash
# GPU box: A/B of LFA-backward occupancy caps (launch bounds) and the software-pipelined variant, same box
set -u
TAG=${1:-bwd}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/eval_$TAG.log; : > $LOG
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
run() { echo "=== $*" >> $LOG; ( eval "$@" ) >> $LOG 2>&1; echo "rc=$?" >> $LOG; }
T='python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_train.py -q -x -k "lfa or net or train or oracle or golden or full" 2>&1 | tail -3'
run "timeout -s KILL 60 python tools/opbench.py lfa 2>&1 | grep '^lfa'"
run "M3D_LFA_BWD_PIPE=1 timeout -s KILL 150 $T"
run "M3D_LFA_BWD_PIPE=1 timeout -s KILL 60 python tools/opbench.py lfa 2>&1 | grep '^lfa'"
for n in w64 w32 w128 wall; do
  run "M3D_LIB=$V/libm3d_bwd_$n.so timeout -s KILL 60 python tools/opbench.py lfa 2>&1 | grep '^lfa'"
done
run "M3D_LIB=$V/libm3d_bwd_wall.so M3D_LFA_BWD_PIPE=1 timeout -s KILL 60 python tools/opbench.py lfa 2>&1 | grep '^lfa'"
run "M3D_LIB=$V/libm3d_bwd_wall.so timeout -s KILL 150 $T"
run "M3D_LIB=$V/libm3d_bwd_wall.so M3D_LFA_BWD_PIPE=1 timeout -s KILL 150 $T"
cat $LOG | cut -c1-200
