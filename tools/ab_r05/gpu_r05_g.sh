#!/bin/bash
# GPU box, round 5 call g: (1) SQ counters of the roofline kernels (level-1 LFA backward: where do its cycles go?), (2) kernel
# trace of the predict chain: per-queue busy time, overlap of the two streams, gaps on the main queue
set -u
TAG=${1:-r05g}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
bash tools/gpu_pmc.sh ${TAG}_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" python tools/pmc_target.py | grep -E "lfa_|knn_query" | cut -c1-330
bash tools/gpu_pmc.sh ${TAG}_sq2 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT" python tools/pmc_target.py | grep -E "lfa_|knn_query" | cut -c1-330
rm -rf /tmp/prof && mkdir -p /tmp/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o trace -- python $ROOT/tools/predict_trace.py ) > $OUT/predict_trace_$TAG.log 2>&1
f=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee $OUT/predict_trace_summary_$TAG.log
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "")) for r in rows)
# the LAST predict_cloud call: from the last tile_sel_min_kernel on
starts = [i for i, e in enumerate(ev) if "tile_sel_min_kernel" in e[2]]
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
for q, iv in perq.items():
    print(f"queue {q}: n={len(iv)} busy {sum(e - s for s, e, _ in iv) / 1e6:.2f} ms, first {(iv[0][0] - t0) / 1e6:.2f} last end {(max(e for _, e, _ in iv) - t0) / 1e6:.2f}")
    agg = collections.Counter()
    for s, e, n in iv: agg[n] += e - s
    for n, v in agg.most_common(8): print(f"      {n:60s} {v / 1e6:7.2f} ms")
mainq = max(perq, key=lambda q: sum(e - s for s, e, _ in perq[q]))
iv = sorted(perq[mainq]); gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(iv, iv[1:]):
    if s1 - e0 > 100000: gaps.append((s1 - e0, (e0 - t0) / 1e6, n0, n1))
print(f"main queue {mainq}: gaps > 0.1 ms: {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)}")
for g in sorted(gaps, reverse=True)[:16]: print(f"   gap {g[0] / 1e6:6.2f} ms at t={g[1]:6.2f}: after {g[2]} -> before {g[3]}")
PY
