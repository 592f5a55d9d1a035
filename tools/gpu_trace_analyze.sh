#!/bin/bash
# kernel-trace of bench.py (graph mode, train steps only) + interval analysis: GPU-busy union, per-queue busy, gaps
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-t}
EXTRA=${2:-}
export OUT TAG
rm -rf /tmp/prof && mkdir -p /tmp/prof
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --skip-cpu-baseline --skip-roofline --skip-extras $EXTRA ) > $OUT/trace_$TAG.log 2>&1
f=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Grid_Size_X", r.get("Grid_Size", ""))) for r in rows))
# find adam kernels = step boundaries
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
print("adam launches", len(adam))
if len(adam) >= 4:
    a, b = adam[-3], adam[-2]   # one full train step between two Adam launches (timed region)
    seg = ev[a + 1:b + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    import os
    with open(os.path.join(os.environ.get("OUT", "."), "step_timeline_" + os.environ.get("TAG", "t") + ".csv"), "w") as fh:
        fh.write("start_us,dur_us,queue,grid_x,kernel\n")
        for s_, e_, n_, q_, g_ in seg:
            short = n_.split('(')[0].replace('void ', '').replace(',', ';').replace(' ', '')[:100]
            if "at::native" in n_:  # torch kernels: the functor says what it is (the template head does not)
                import re
                m_ = re.search(r"(CUDAFunctor\w+<[\w ]+>|FillFunctor<[\w ]+>|fused_dropout\w+|masked_scale\w+|\w+Functor\w*<[\w ]+>|normal_kernel|CatArray\w+)", n_)
                short = "torch:" + (m_.group(1) if m_ else n_[:70].replace(",", ";"))
            fh.write(f"{(s_ - t0) / 1e3:.1f},{(e_ - s_) / 1e3:.1f},{q_},{g_},{short}\n")
    wall = (t1 - t0) / 1e3
    # union of intervals
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, *_ in seg[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, *_ in seg)
    print(f"step: {len(seg)} kernels, wall {wall:.1f} us, union-busy {busy/1e3:.1f} us, sum of durations {tot/1e3:.1f} us, idle {wall - busy/1e3:.1f} us")
    perq = collections.Counter()
    for s, e, n, q, _g in seg: perq[q] += e - s
    print("per queue busy us:", {q: round(v / 1e3, 1) for q, v in perq.items()})
    agg = collections.Counter(); cnt = collections.Counter()
    for s, e, n, q, _g in seg:
        k = n.split("(")[0][:48]; agg[k] += e - s; cnt[k] += 1
    for k, v in agg.most_common(12): print(f"  {k:50s} n={cnt[k]:4d} {v/1e3:8.1f} us")
    # gaps on the busiest queue (main stream): where does it wait?
    mainq = perq.most_common(1)[0][0]
    mq = [e for e in seg if e[3] == mainq]
    print("main queue", mainq, "kernels", len(mq), "span us", (mq[-1][1] - mq[0][0]) / 1e3, "first start rel", (mq[0][0] - t0) / 1e3)
    gaps = []
    for (s0, e0, n0, *_a), (s1, e1, n1, *_b) in zip(mq, mq[1:]):
        if s1 - e0 > 15000: gaps.append((s1 - e0, (e0 - t0) / 1e3, n0.split("(")[0][:40], n1.split("(")[0][:40]))
    print("main-queue gaps > 15 us: total", sum(g[0] for g in gaps) / 1e3, "us in", len(gaps))
    for g in sorted(gaps, reverse=True)[:25]: print(f"   gap {g[0]/1e3:7.1f} us at t={g[1]:7.1f}: after {g[2]} -> before {g[3]}")
    for q in perq:
        qq = [e for e in seg if e[3] == q]
        print(f"queue {q}: n={len(qq)} first {(qq[0][0]-t0)/1e3:.1f} last end {(qq[-1][1]-t0)/1e3:.1f} busy {perq[q]/1e3:.1f}")
PY
