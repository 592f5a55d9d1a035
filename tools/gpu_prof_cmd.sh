#!/bin/bash
# rocprofv3 kernel-trace of an arbitrary command on the GPU box; prints the per-kernel summary.
# usage: tools/gpu_prof_cmd.sh TAG cmd args...
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
TAG=$1; shift
rm -rf /tmp/prof && mkdir -p /tmp/prof
( cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- "$@" ) > $OUT/rocprofcmd_$TAG.log 2>&1
for f in $(find /tmp/prof -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats_$TAG.csv; done
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats_$TAG.csv")))
for r in rows[:25]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1000:9.2f} min_us={float(r['MinNs'])/1000:8.2f} max_us={float(r['MaxNs'])/1000:9.2f} pct={r['Percentage']}")
PY
