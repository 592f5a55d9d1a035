#!/bin/bash
# GPU box, round 5 call e: the predict chain after its rework (lane-mask tile selection, preparation of batch b + 1 on a side
# stream, no per-batch host round trips on the main stream): parity tests, the end-to-end leg, the stage profile
set -u
TAG=${1:-r05e}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout -s KILL 400 python -m pytest tests/test_tiling.py tests/test_gpu_predict.py tests/test_gpu_prep.py -m gpu -x -q --timeout 300 2>&1 | grep -v "^  File\|^Extension modules\|amdgpu.ids" | tail -8 | cut -c1-300 | tee $OUT/pytest_predict_$TAG.log
timeout -s KILL 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/predict_e2e_$TAG.log
import argparse, json, torch, bench
args = argparse.Namespace()
dev = torch.device("cuda:0")
for rep in range(2):
    print(json.dumps(bench.predict_e2e_bench(args, dev, reps=3)))
PY
