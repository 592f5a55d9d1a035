#!/bin/bash
# GPU box, round 4 call D: side-stream choice by measurement (N > 1 launch form), bucket-skipping FPS parity + timing, host
# time of the variable-layout step with single-threaded autograd.
set -u
TAG=${1:-r04d}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 300 python -m pytest tests/test_gpu_pointnet2.py tests/test_gpu_train.py -m gpu -q --timeout 300 -x 2>&1 | grep -v "^  File\|^Extension modules" | tail -15 > $OUT/pytest_gpu_$TAG.log; tail -5 $OUT/pytest_gpu_$TAG.log | cut -c1-240
for c in none pg captured eager; do timeout -s KILL 150 python tools/collective_probe.py $c 2>&1 | grep collective_probe; done > $OUT/collective_probe_$TAG.log; cat $OUT/collective_probe_$TAG.log
timeout -s KILL 200 python bench.py --mode pointnet2 --steps 3 --tiles 16 --points 40000 --neighbors 32 2>/dev/null | tail -1 > $OUT/pointnet2_$TAG.json; cut -c1-420 $OUT/pointnet2_$TAG.json; echo
timeout -s KILL 200 python bench.py --mode pointnet2 --steps 3 --tiles 16 --points 12800 --neighbors 16 2>/dev/null | tail -1 > $OUT/pointnet2_12800_$TAG.json; cut -c1-420 $OUT/pointnet2_12800_$TAG.json; echo
for m in "variable" "variable st" "st"; do echo "== host_profile $m"; timeout -s KILL 200 python tools/host_profile.py $m 2>&1 | grep -E "^host|run_backward|tolist" ; done | tee $OUT/host_profile_$TAG.log
