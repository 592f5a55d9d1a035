#!/bin/bash
set -u
TAG=${1:-r02m}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/lfa_stagger_$TAG.log; : > $L
for sgr in 0 1 2 3 5 8; do
  echo "=== M3D_LFA_BWD_STAGGER=$sgr" >> $L
  M3D_LFA_BWD_STAGGER=$sgr timeout -s KILL 200 python tools/opbench.py lfa 2>&1 | grep -i "lfa level" | cut -c1-140 >> $L
done
cat $L
