#!/bin/bash
# GPU box, round 5 call b: the whole GPU suite (with the parity margins printed), then the phase ablation of the FULL LFA
# backward kernels (LFA_BWD_DBG variants built by the caller into myria3d_amd/variants/libm3d_bwd_dbg{2,4,8,16,32}.so)
set -u
TAG=${1:-r05b}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout -s KILL 1100 python -m pytest tests -m gpu -q -s --timeout 400 --durations=8 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" > $OUT/pytest_full_$TAG.log
grep -E "passed|failed|error" $OUT/pytest_full_$TAG.log | tail -3 | cut -c1-250
grep -E "FAILED|Error|assert" $OUT/pytest_full_$TAG.log | head -20 | cut -c1-300
grep -E "^\[parity\] .*(logits|log_probas)" $OUT/pytest_full_$TAG.log | cut -c1-220 > $OUT/parity_margins_$TAG.log; cat $OUT/parity_margins_$TAG.log | head -40
for v in 2 4 8 16 32; do
  L=$ROOT/myria3d_amd/variants/libm3d_bwd_dbg$v.so
  [ -f $L ] && { echo "== LFA_BWD_DBG=$v (stop after phase: 2=1, 4=2, 8=3, 16=5, 32=6)"; M3D_LIB=$L timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa" | sed 's/fwd.*bwd/bwd/'; }
done > $OUT/lfa_bwd_phases_$TAG.log 2>&1; cat $OUT/lfa_bwd_phases_$TAG.log
{ echo "== product"; timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa" | sed 's/fwd.*bwd/bwd/'
  echo "== no dx atomics, pipelined kernels kept (LFA_BWD_DBG=1)"; M3D_LIB=$ROOT/myria3d_amd/variants/libm3d_noatom.so timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa" | sed 's/fwd.*bwd/bwd/'; } > $OUT/lfa_bwd_noatom_$TAG.log 2>&1; cat $OUT/lfa_bwd_noatom_$TAG.log
