#!/bin/bash
# rocprofv3 PMC pass (counters only, no other tracing) of a command on the GPU box; prints per-kernel counter means.
# usage: tools/gpu_pmc.sh TAG "COUNTER1 COUNTER2 ..." cmd args...
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
TAG=$1; CTRS=$2; shift; shift
rm -rf /tmp/pmc && mkdir -p /tmp/pmc
( cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --pmc $CTRS --output-format csv -d /tmp/pmc -o pmc -- "$@" ) > $OUT/pmc_$TAG.log 2>&1
f=$(find /tmp/pmc -name '*counter_collection.csv' | head -1)
[ -z "$f" ] && { tail -20 $OUT/pmc_$TAG.log; exit 1; }
python - "$f" "$OUT/pmc_$TAG.csv" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    # one row per kernel NAME and launch geometry: the same template runs at four level sizes inside a step
    k = (r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Counter_Name"])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
kern = collections.OrderedDict()
for (kn, cn), (n, v) in agg.items():
    kern.setdefault(kn, {})[cn] = v / n
    kern[kn]["_calls"] = n
w = csv.writer(open(sys.argv[2], "w"))
names = sorted({c for d in kern.values() for c in d})
w.writerow(["kernel"] + names)
for kn, d in kern.items():
    w.writerow([kn] + [f"{d.get(c, 0):.4g}" for c in names])
    print(kn, " ".join(f"{c}={d.get(c,0):.4g}" for c in names))
PY
