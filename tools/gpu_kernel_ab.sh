#!/bin/bash
# GPU box: per-kernel average durations of the eager training step (rocprofv3 --kernel-trace --stats) for several variant
# libraries on one box.  usage: gpu_kernel_ab.sh TAG "old stock ..." "wgrad|adam" [bench args]
set -u
export TMPDIR=/tmp
TAG=$1; LIBS=$2; PAT=$3; shift; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; : > $OUT/kernel_ab_$TAG.log
for L in $LIBS; do
  if [ $L = stock ]; then unset M3D_LIB; else export M3D_LIB=$GRAFT_REPO_ROOT/myria3d_amd/variants/libm3d_$L.so; fi
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 8 --warmup 2 --skip-cpu-baseline --skip-extras --skip-roofline "$@" ) > $OUT/kernel_ab_run.log 2>&1
  f=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
  python - "$f" "$L" "$PAT" <<'PY' | tee -a $OUT/kernel_ab_$TAG.log
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0])
    if not re.search(sys.argv[3], n): continue
    k = (n, r.get("Grid_Size_X", r.get("Grid_Size", "")))
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = 0.0
for (n, g), a in agg.items():
    print(f"{sys.argv[2]:8s} {n[:60]:60s} grid {g:>9s} x{a[0]:3d} avg {a[1] / a[0] / 1e3:8.1f} us"); tot += a[1] / a[0] / 1e3 * (a[0] // max(1, min(x[0] for x in agg.values())))
print(f"{sys.argv[2]:8s} sum of averages (per step) {sum(a[1] for a in agg.values()) / 1e3 / max(1, min(x[0] for x in agg.values())):8.1f} us")
PY
done
