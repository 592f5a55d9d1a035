#!/bin/bash
set -u
TAG=${1:-r02o}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/lfa_fwd_pipe_$TAG.log; : > $L
timeout -s KILL 600 python -m pytest tests -m gpu -q -x -k "lfa or eval_logits or golden or train_forward or baseline_tiles or dense" 2>&1 | tail -3 >> $L
run() { echo "=== $1" >> $L; env $2 timeout -s KILL 200 python tools/opbench.py lfa 2>&1 | grep -i "lfa level [12]" | cut -c1-75 >> $L; }
run "one-shot kernel" "M3D_LFA_FWD_PIPE=0"
run "pipe cap 1024" "M3D_LFA_FWD_CAP=1024"
run "pipe cap 768" "M3D_LFA_FWD_CAP=768"
run "pipe cap 1536" "M3D_LFA_FWD_CAP=1536"
run "pipe cap 2048" "M3D_LFA_FWD_CAP=2048"
run "pipe minw5 cap 1280" "M3D_LFA_FWD_CAP=1280 M3D_LIB=$PWD/myria3d_amd/variants/libm3d_fwd_m5.so"
run "pipe minw6 cap 1536" "M3D_LFA_FWD_CAP=1536 M3D_LIB=$PWD/myria3d_amd/variants/libm3d_fwd_m6.so"
cat $L
