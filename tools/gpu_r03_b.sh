#!/bin/bash
# GPU box, round 3 second call: the new parity tests only (per-test timeout), the staged-kNN sweep, then the bench legs
# one by one under tight timeouts (which one hung in r03a?) and the full line.  usage: tools/gpu_r03_b.sh TAG
set -u
TAG=${1:-r03b}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
T=$OUT/pytest_gpu_$TAG.log
timeout -s KILL 420 python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --timeout 150 \
  -k "graphed or stale or collective or shared_input or knn or batched or reference_fixture or flattened or train_steps_follow or hipgraph" 2>&1 | tail -40 > $T
grep -E "passed|failed|FAILED|Error|Timeout|\[parity\] (Graphed|1-rank)" $T | tail -20
bash tools/gpu_r03_knn.sh $TAG > /dev/null 2>&1
grep -E "^knn_bench|passed|failed" $OUT/knn_staged_$TAG.log | cut -c1-220
leg() { name=$1; shift; /usr/bin/time -f "$name: %e s" timeout -s KILL "$@" > $OUT/leg_${name}_$TAG.json 2> $OUT/leg_${name}_$TAG.err; echo "rc=$? $(tail -1 $OUT/leg_${name}_$TAG.err)"; tail -c 600 $OUT/leg_${name}_$TAG.json; echo; }
leg collective 150 python bench.py --force-collective --skip-cpu-baseline --skip-roofline --skip-extras
leg dropin 120 python bench.py --mode dropin
SKIP=""
grep -q allreduce_ms $OUT/leg_collective_$TAG.json || SKIP="collective"
/usr/bin/time -f "full line: %e s" timeout -s KILL 420 python bench.py --skip-legs "$SKIP" 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
grep -E "^\[bench|full line" $OUT/bench_$TAG.err | tail -30
cut -c1-1200 $OUT/bench_$TAG.json
