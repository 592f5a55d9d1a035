#!/bin/bash
# GPU box: kernel trace of ONE predict_cloud call (tools/predict_trace.py: bench.py's 10 M-point cloud, after a warm-up call):
# per-queue busy time, the kernels that fill each queue, the gaps of the main queue.  Usage: gpu_predict_trace.sh TAG
set -u
TAG=${1:-r06x}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
rm -rf /tmp/prof && mkdir -p /tmp/prof
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o trace -- python $ROOT/tools/predict_trace.py ) > $OUT/predict_trace_$TAG.log 2>&1
tail -2 $OUT/predict_trace_$TAG.log | cut -c1-400
f=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee $OUT/predict_trace_summary_$TAG.log
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r"^void ", "", n.split("(")[0])
    return n[:64]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "")) for r in rows)
starts = [i for i, e in enumerate(ev) if "tile_sel_min_kernel" in e[2]]  # the LAST predict_cloud call
seg = ev[starts[-1]:]
t0, t1 = seg[0][0], max(e[1] for e in seg)
print(f"last predict_cloud call: {len(seg)} kernels, wall {(t1 - t0) / 1e6:.2f} ms")
perq = collections.defaultdict(list)
for s, e, n, q in seg: perq[q].append((s, e, n))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0][0], iv[0][1]
    for s, e, *_ in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
print(f"union busy (any queue) {union([(s, e) for s, e, *_ in seg]) / 1e6:.2f} ms")
for q, iv in sorted(perq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    print(f"queue {q}: n={len(iv)} busy {sum(e - s for s, e, _ in iv) / 1e6:.2f} ms, first {(iv[0][0] - t0) / 1e6:.2f} last end {(max(e for _, e, _ in iv) - t0) / 1e6:.2f}")
    agg = collections.Counter(); cnt = collections.Counter()
    for s, e, n in iv: agg[n] += e - s; cnt[n] += 1
    for n, v in agg.most_common(14): print(f"      {n:64s} {cnt[n]:4d} x {v / 1e6:7.2f} ms")
mainq = max(perq, key=lambda q: sum(e - s for s, e, _ in perq[q]))
iv = sorted(perq[mainq]); gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(iv, iv[1:]):
    if s1 - e0 > 50000: gaps.append((s1 - e0, (e0 - t0) / 1e6, n0, n1))
print(f"main queue {mainq}: gaps > 0.05 ms: {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)}")
for g in sorted(gaps, reverse=True)[:20]: print(f"   gap {g[0] / 1e6:6.2f} ms at t={g[1]:6.2f}: after {g[2]} -> before {g[3]}")
# one batch in the middle of the call, kernel by kernel on the main queue (from its fc0 GEMM to the next one)
names = [n for _, _, n in iv]
PY
