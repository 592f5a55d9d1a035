#!/bin/bash
# GPU box, round 5 call f: workgroup geometry of the level-1 LFA backward (ch 8 / 16) — one or two waves per workgroup
# (barriers between the phases of a group become trivial) against the stock 4 waves x 128 rows; + the predict chain again
set -u
TAG=${1:-r05f}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
{ echo "== stock"; timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa level 1"
for v in w1 w1b w2 w2b; do echo "== $v"; M3D_LIB=$ROOT/myria3d_amd/variants/libm3d_$v.so timeout -s KILL 150 python tools/opbench.py lfa | grep "^lfa level 1"; done; } 2>&1 | grep -v amdgpu.ids | tee $OUT/lfa_bwd_l1_geometry_$TAG.log
for v in w1 w2; do M3D_LIB=$ROOT/myria3d_amd/variants/libm3d_$v.so timeout -s KILL 200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q --timeout 200 -k "lfa_train_full or persistent" 2>&1 | grep -E "passed|failed" | tail -1; done
timeout -s KILL 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/predict_e2e_$TAG.log
import argparse, json, torch, bench
print(json.dumps(bench.predict_e2e_bench(argparse.Namespace(), torch.device("cuda:0"), reps=3)))
PY
