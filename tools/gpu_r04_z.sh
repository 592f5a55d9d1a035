#!/bin/bash
# GPU box, round 4 call Z (the last 2 GPU-minutes): the new colsum / cross-entropy parity cases, and rows per thread of the
# BatchNorm column sums now that a trip holds eight rows (BN_BWD_RPT 2 / 4 against the default 8 / 16), per layer and in the step.
set -u
TAG=${1:-r04z2}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/myria3d_amd/variants
timeout -s KILL 100 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -m gpu -x -q -k "colsum_row_shapes or cross_entropy" 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $OUT/pytest_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms')"; }
for v in default rpt2 rpt4; do
  L=""; [ $v != default ] && L=$V/libm3d_$v.so
  echo "== $v"
  M3D_LIB=$L timeout -s KILL 60 python tools/opbench.py bnbwd 2>&1 | grep -E "^b1\.|^cls|^fp1|^b2\.mlp1|^b4\.mlp1|TOTAL" | cut -c1-62
  M3D_LIB=$L timeout -s KILL 60 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "step $v"
done 2>&1 | tee $OUT/bn_rpt_$TAG.log
