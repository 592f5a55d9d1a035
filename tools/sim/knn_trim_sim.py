"""Lock-step cost of the per-lane ring walk with and without circular trimming of a ring's runs (exact run structure of
knn_query_queue_kernel: per ring R, rows dy=-R..R; edge rows are one run over cx-R..cx+R, inner rows two single-cell runs)."""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myria3d_amd.synthetic import synthetic_tile
from scipy.spatial import cKDTree

def run(tid, cell_target, K=16, trim=True, gmax=64, batch=8):
    _, pos, _ = synthetic_tile(12800, tid)
    pos = pos.numpy().astype(np.float64)
    n = len(pos)
    x, y = pos[:, 0], pos[:, 1]
    gx0, gy0 = x.min(), y.min()
    wx, wy = x.max() - gx0, y.max() - gy0
    wmax = max(wx, wy)
    h = max(math.sqrt(wx * wy * cell_target / n), wmax / gmax * 1.0001)
    Gx, Gy = min(gmax, int(wx / h) + 1), min(gmax, int(wy / h) + 1)
    cx = np.clip(((x - gx0) / h).astype(int), 0, Gx - 1); cy = np.clip(((y - gy0) / h).astype(int), 0, Gy - 1)
    cell = cy * Gx + cx
    order = np.argsort(cell, kind="stable")
    cs = np.zeros(Gx * Gy + 1, int); np.add.at(cs, cell + 1, 1); cs = np.cumsum(cs)
    spos = pos[order]
    tot_trips = 0; tot_runs = 0; tot_cand = 0; waves = 0
    for w0 in range(0, n, 64):
        q = order[w0:w0 + 64]
        L = len(q)
        best = [np.full(0, np.inf) for _ in range(L)]  # sorted d2 lists
        kth = np.full(L, np.inf)
        active = np.ones(L, bool)
        R = 0
        while active.any():
            # runs of this ring: list of (dy, xa_off, xb_off)
            runs = []
            for dy in range(-R, R + 1):
                if abs(dy) == R: runs.append((dy, -R, R))
                else: runs.append((dy, -R, -R)); runs.append((dy, R, R))
            newc = [[] for _ in range(L)]
            for (dy, xa, xb) in runs:
                lens = np.zeros(L, int)
                for li in range(L):
                    if not active[li]: continue
                    qi = q[li]; yy = cy[qi] + dy
                    if yy < 0 or yy >= Gy: continue
                    a = max(cx[qi] + xa, 0); b = min(cx[qi] + xb, Gx - 1)
                    if a > b: continue
                    if trim and np.isfinite(kth[li]):
                        # gap in y to this row band
                        y0r = gy0 + yy * h; y1r = y0r + h
                        gy = 0.0 if y0r <= y[qi] <= y1r else min(abs(y[qi] - y0r), abs(y[qi] - y1r))
                        rem = kth[li] - gy * gy
                        if rem < 0: continue
                        xr = math.sqrt(rem)
                        a = max(a, int(math.floor((x[qi] - xr - gx0) / h))); b = min(b, int(math.floor((x[qi] + xr - gx0) / h)))
                        if a > b: continue
                    p0, p1 = cs[yy * Gx + a], cs[yy * Gx + b + 1]
                    lens[li] = p1 - p0
                    if p1 > p0:
                        d = spos[p0:p1] - pos[qi]
                        newc[li].append((d * d).sum(1))
                m = lens.max()
                if m > 0:
                    tot_trips += math.ceil(m / batch); tot_runs += 1
                tot_cand += lens.sum()
            # ring end: update lists, termination
            for li in range(L):
                if not active[li]: continue
                if newc[li]:
                    allc = np.sort(np.concatenate([best[li]] + newc[li]))[:K]
                    best[li] = allc
                    if len(allc) >= K: kth[li] = allc[K - 1]
                qi = q[li]
                covers = (cx[qi] - R <= 0) and (cx[qi] + R >= Gx - 1) and (cy[qi] - R <= 0) and (cy[qi] + R >= Gy - 1)
                if covers: active[li] = False; continue
                b = 3.4e38
                if cx[qi] - R > 0: b = min(b, x[qi] - (gx0 + (cx[qi] - R) * h))
                if cx[qi] + R < Gx - 1: b = min(b, (gx0 + (cx[qi] + R + 1) * h) - x[qi])
                if cy[qi] - R > 0: b = min(b, y[qi] - (gy0 + (cy[qi] - R) * h))
                if cy[qi] + R < Gy - 1: b = min(b, (gy0 + (cy[qi] + R + 1) * h) - y[qi])
                if kth[li] <= b * b: active[li] = False
            R += 1
        waves += 1
    return tot_trips / waves, tot_runs / waves, tot_cand / n

for ct in (7.0, 4.0):
    for trim in (False, True):
        t, r, c = run(0, ct, trim=trim)
        print(f"cell_target {ct} trim={trim}: trips(8/trip) per wave {t:.0f} (= {8*t:.0f} slots), runs/wave {r:.0f}, cand/query {c:.0f}")
