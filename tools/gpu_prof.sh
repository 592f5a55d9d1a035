#!/bin/bash
# rocprofv3 kernel-trace of bench.py on the GPU box (via gpurun). usage: tools/gpu_prof.sh TAG [bench args...]
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
TAG=${1:-r01}; shift
rm -rf /tmp/prof && mkdir -p /tmp/prof
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 5 --warmup 2 --skip-cpu-baseline --skip-extras "$@" ) > $OUT/rocprof_$TAG.log 2>&1
find /tmp/prof -type f | head -20
for f in $(find /tmp/prof -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats_$TAG.csv; done
for f in $(find /tmp/prof -name '*kernel_trace.csv'); do python - "$f" "$OUT/kernel_trace_${TAG}_summary.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["kernel", "grid", "wg", "vgpr", "lds", "calls", "total_ns", "avg_ns", "pct"])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow(list(k) + [a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 2)])
PY
done
head -50 $OUT/kernel_stats_$TAG.csv
