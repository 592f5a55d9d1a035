#!/bin/bash
# GPU box, round 5 call a: parity of the mask-free LFA kernels (M3D_LFA_FULL), their per-level times against the general
# kernels, the step with / without them; the two A/Bs still pending from round 4 (LFA_RED_WIDE, ROWS_GATHER_BATCH); the price
# of the dx atomics per level on the new kernels (LFA_BWD_DBG=1 variant).   usage: tools/gpu_r05_a.sh [TAG]
set -u
TAG=${1:-r05a}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
V=$ROOT/myria3d_amd/variants
timeout -s KILL 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -x -q --timeout 300 -k "lfa or golden or reference or eval_logits or flattened" 2>&1 | grep -v "^  File\|^Extension modules" | tail -8 > $OUT/pytest_$TAG.log; tail -3 $OUT/pytest_$TAG.log | cut -c1-250
{ echo "== full (round 5 kernels)"; timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa"
  echo "== general kernels (nofull)"; timeout -s KILL 150 python tools/opbench.py lfa nofull | grep "^lfa"
  echo "== full, no dx atomics (LFA_BWD_DBG=1: WRONG dx, timing only)"; M3D_LIB=$V/libm3d_noatom.so timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa"
} > $OUT/lfa_opbench_$TAG.log 2>&1; cat $OUT/lfa_opbench_$TAG.log
step() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'ms; eval fwd', d['fwd_only']['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for rep in 1 2; do
timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph full   "
M3D_LFA_FULL=0 timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph general"
M3D_LIB=$V/libm3d_pend.so timeout -s KILL 200 python bench.py --skip-extras --skip-cpu-baseline --launch graph 2>/dev/null | tail -1 | step "graph full + RED_WIDE + GATHER_BATCH"
done 2>&1 | tee $OUT/step_$TAG.log
M3D_LIB=$V/libm3d_pend.so timeout -s KILL 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -m gpu -x -q --timeout 200 -k "csr or gather or persistent or graphed" 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $OUT/pytest_pend_$TAG.log
