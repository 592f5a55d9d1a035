#!/bin/bash
# GPU box: same-box A/B of the kNN query variants + SQ counters of the level-1 query.  usage: tools/gpu_r02_knn2.sh TAG
set -u
TAG=${1:-r02c}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
L=$OUT/knn_ab_$TAG.log; : > $L
timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
M3D_KNN_QUEUE=0 timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
M3D_KNN_KEYS=u64 timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
for v in myria3d_amd/variants/libm3d_knn_*.so; do
  M3D_LIB=$PWD/$v timeout -s KILL 120 python tools/knn_bench.py 2>&1 | grep knn_bench >> $L
done
cat $L
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
bash tools/gpu_pmc.sh ${TAG}_q1 "$C" python tools/knn_bench.py pmc | grep -i knn_query
M3D_KNN_QUEUE=0 bash tools/gpu_pmc.sh ${TAG}_q0 "$C" python tools/knn_bench.py pmc | grep -i knn_query
C2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
bash tools/gpu_pmc.sh ${TAG}_q1b "$C2" python tools/knn_bench.py pmc | grep -i knn_query
M3D_KNN_QUEUE=0 bash tools/gpu_pmc.sh ${TAG}_q0b "$C2" python tools/knn_bench.py pmc | grep -i knn_query
