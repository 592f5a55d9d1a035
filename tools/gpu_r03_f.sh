#!/bin/bash
# GPU box: full parity suite, A/B of the train step (gradient slots, lookahead form), the full default bench line with its
# leg-by-leg clock.  usage: tools/gpu_r03_f.sh TAG
set -u
TAG=${1:-r03f}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout -s KILL 600 python -m pytest tests -m gpu -q --timeout 200 --durations=5 2>&1 | tail -14 > $OUT/pytest_gpu_$TAG.log; tail -9 $OUT/pytest_gpu_$TAG.log | cut -c1-200
ab() { name=$1; shift; env "$@" timeout -s KILL 100 python bench.py --skip-cpu-baseline --skip-roofline --skip-extras --launch graph $EXTRA 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], 'ms/step; fwd', d['fwd_only']['ms_per_step'])"; }
: > $OUT/ab_$TAG.log
for rep in 1 2; do
  EXTRA="" ab "default(dual,slots)" M3D_X=1 | tee -a $OUT/ab_$TAG.log
  EXTRA="" ab "no-grad-slots" M3D_GRAD_SLOTS=0 | tee -a $OUT/ab_$TAG.log
  EXTRA="--lookahead-mode single" ab "single-graph" M3D_X=1 | tee -a $OUT/ab_$TAG.log
done
SECONDS=0
timeout -s KILL 500 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
echo "full line: ${SECONDS}s"; grep -E "^\[bench" $OUT/bench_$TAG.err | tail -25; cut -c1-600 $OUT/bench_$TAG.json
