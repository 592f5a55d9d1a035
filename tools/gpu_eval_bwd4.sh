#!/bin/bash
# GPU box: last A/B of round 1: 16-wave workgroup for ch = 256, five workgroups per CU for ch <= 32
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/eval_r01u.log; : > $LOG
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
run() { echo "=== $*" >> $LOG; ( eval "$@" ) >> $LOG 2>&1; echo "rc=$?" >> $LOG; }
run "M3D_LIB=$V/libm3d_bwd_v256.so timeout -s KILL 25 python tools/opbench.py lfa 2>&1 | grep '^lfa level 4'"
run "M3D_LIB=$V/libm3d_bwd_v256.so timeout -s KILL 40 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -q -x -k 'lfa or train_forward' 2>&1 | tail -2"
run "M3D_LIB=$V/libm3d_bwd_v16w5.so timeout -s KILL 25 python tools/opbench.py lfa 2>&1 | grep '^lfa level 1'"
run "M3D_LIB=$V/libm3d_bwd_v32w5.so timeout -s KILL 25 python tools/opbench.py lfa 2>&1 | grep '^lfa level 2'"
run "timeout -s KILL 25 python tools/opbench.py lfa 2>&1 | grep '^lfa'"
cat $LOG | cut -c1-160
