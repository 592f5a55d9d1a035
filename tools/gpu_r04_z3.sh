#!/bin/bash
# GPU box, round 4 call Z3: plans built from host-side tile sizes (asynchronous upload, no device read-back in the step): the
# parity test and the drop-in bench with the two new legs.
set -u
TAG=${1:-r04z3}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 60 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "host_side_tile_sizes or changing_tile_layouts or stale_prefetch" 2>&1 | grep -E "passed|failed|rror|assert" | tail -6 | tee $OUT/pytest_$TAG.log
timeout -s KILL 60 python bench.py --mode dropin 2>$OUT/dropin_$TAG.err | tail -1 > $OUT/dropin_$TAG.json; cut -c1-900 $OUT/dropin_$TAG.json; tail -3 $OUT/dropin_$TAG.err
