#!/bin/bash
# GPU box: what the train-mode statistics epilogue of the forward GEMMs costs (GEMM_DBG variants: WRONG statistics, timing only)
for v in "" gdbg1 gdbg2 gdbg4; do
  echo "== variant ${v:-stock}"
  if [ -n "$v" ]; then export M3D_LIB=$GRAFT_REPO_ROOT/myria3d_amd/variants/libm3d_$v.so; else unset M3D_LIB; fi
  timeout -s KILL 120 python tools/opbench.py gemm 2>&1 | grep -E "b1.short|cls1|b2.post2|b3.post2|b3.mlp2|b4.post2|b4.mlp2|fp4|summit|TOTAL" | cut -c1-95
done
