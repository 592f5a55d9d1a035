#!/bin/bash
set -u
TAG=${1:-r02d}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_ab_$TAG.log; : > $L
timeout -s KILL 300 python -m pytest tests -m gpu -q -x -k "knn or eval_logits or full_size or seeded or golden or interpolate" 2>&1 | tail -3 >> $L
timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
M3D_KNN_QUEUE=0 timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
for v in myria3d_amd/variants/libm3d_knn_*.so; do
  M3D_LIB=$PWD/$v timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
done
cat $L
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
bash tools/gpu_pmc.sh ${TAG}_q1 "$C" python tools/knn_bench.py pmc | grep -i knn_query
L2=$OUT/lfa_bwd_phases_$TAG.log; : > $L2
for d in 0 1 2 4 8 16 32; do
  echo "=== M3D_LFA_BWD_DBG=$d" >> $L2
  M3D_LFA_BWD_DBG=$d timeout -s KILL 200 python tools/opbench.py lfa 2>&1 | grep -i "lfa" >> $L2
done
cat $L2
echo "=== M3D_LFA_BWD_PIPE=0 (non-pipelined, all phases)" >> $L2
M3D_LFA_BWD_PIPE=0 timeout -s KILL 200 python tools/opbench.py lfa 2>&1 | grep -i "lfa" >> $L2
tail -9 $L2
