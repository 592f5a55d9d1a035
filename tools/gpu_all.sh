#!/bin/bash
# GPU box: parity tests, bench line, rocprofv3 kernel trace.  usage: tools/gpu_all.sh TAG
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout 240 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest_gpu_$TAG.log
cat $OUT/pytest_gpu_$TAG.log | tail -6
timeout 200 python bench.py 2> $OUT/bench_$TAG.err | tail -1 > $OUT/bench_$TAG.json
cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
bash tools/gpu_prof.sh $TAG > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.reader(open("$OUT/kernel_trace_${TAG}_summary.csv")))
for r in rows[1:45]:
    print(f"{r[0][:56]:56s} g={r[1]:>8s} calls={r[5]:>5s} tot_us={int(r[6])/1000:9.1f} avg_us={float(r[7])/1000:8.1f} pct={r[8]}")
PY
