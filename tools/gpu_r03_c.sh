#!/bin/bash
# GPU box, round 3 third call: GraphedStep parity (dual graph captured on the side stream), kNN ring histogram + per-stage
# kernel durations, bench legs one by one.  usage: tools/gpu_r03_c.sh TAG
set -u
export TMPDIR=/tmp
TAG=${1:-r03c}
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
T=$OUT/pytest_gpu_$TAG.log
timeout -s KILL 300 python -m pytest tests/test_gpu_train.py -m gpu -q --timeout 150 -k "graphed or collective or shared_input" 2>&1 | tail -30 > $T
grep -E "passed|failed|FAILED|Error|Timeout|\[parity\] (Graphed|1-rank)|^E  " $T | head -20
rm -rf /tmp/kh && mkdir -p /tmp/kh
( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kh -o kh -- python $GRAFT_REPO_ROOT/tools/knn_stage_hist.py ) > $OUT/knn_hist_$TAG.log 2>&1
grep -E "ring|survivors|level 1" $OUT/knn_hist_$TAG.log
f=$(find /tmp/kh -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python - "$f" <<'PY' | tee $OUT/knn_stage_kernels_$TAG.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "knn_stage_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("stage kernels in launch order (template args: KMAX, keys, queue depth, G, FIRST, LAST): duration us")
for r in rows:
    name = r["Kernel_Name"].split("knn_stage_kernel")[1].split(">")[0]
    print(f"  <{name}>  grid={r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s}  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}")
PY
leg() { name=$1; shift; SECONDS=0; timeout -s KILL "$@" > $OUT/leg_${name}_$TAG.json 2> $OUT/leg_${name}_$TAG.err; echo "$name: rc=$? ${SECONDS}s $(grep -E '^\[bench' $OUT/leg_${name}_$TAG.err | tail -1)"; tail -c 500 $OUT/leg_${name}_$TAG.json; echo; }
leg collective 150 python bench.py --force-collective --skip-cpu-baseline --skip-roofline --skip-extras
leg dropin 120 python bench.py --mode dropin
leg main 200 python bench.py --skip-cpu-baseline --skip-legs predict,bf16,dropin,collective,torch,dense
leg torch 200 python bench.py --skip-cpu-baseline --skip-roofline --skip-legs predict,bf16,dropin,collective,dense
grep -E "^\[bench" $OUT/leg_main_$TAG.err $OUT/leg_torch_$TAG.err | tail -20
