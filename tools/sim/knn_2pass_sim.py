"""Two-pass per-lane search: pass 1 = the 3x3 cell block of the query (uniform), rho = k-th distance found there (upper
bound of the true one); pass 2 = every cell that intersects the disc of radius rho (trimmed rows), consumed as a FLAT per-lane
candidate stream (no lock-step per run).  Cost model: wave slots = max over lanes."""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myria3d_amd.synthetic import synthetic_tile

def run(tid, cell_target, K=16, gmax=128, R1=1, order_mode="cell", mid=False):
    _, pos, _ = synthetic_tile(12800, tid)
    pos = pos.numpy().astype(np.float64)
    n = len(pos)
    x, y = pos[:, 0], pos[:, 1]
    gx0, gy0 = x.min(), y.min()
    wx, wy = x.max() - gx0, y.max() - gy0
    h = max(math.sqrt(wx * wy * cell_target / n), max(wx, wy) / gmax * 1.0001)
    Gx, Gy = min(gmax, int(wx / h) + 1), min(gmax, int(wy / h) + 1)
    cx = np.clip(((x - gx0) / h).astype(int), 0, Gx - 1); cy = np.clip(((y - gy0) / h).astype(int), 0, Gy - 1)
    cell = cy * Gx + cx
    order = np.argsort(cell, kind="stable")
    cs = np.zeros(Gx * Gy + 1, int); np.add.at(cs, cell + 1, 1); cs = np.cumsum(cs)
    spos = pos[order]
    p1c = np.zeros(n, int); p2c = np.zeros(n, int); p2runs = np.zeros(n, int); p3c = np.zeros(n, int)
    for qi in range(n):
        # pass 1
        d2s = []
        for yy in range(max(cy[qi] - R1, 0), min(cy[qi] + R1, Gy - 1) + 1):
            a, b = max(cx[qi] - R1, 0), min(cx[qi] + R1, Gx - 1)
            p0, p1 = cs[yy * Gx + a], cs[yy * Gx + b + 1]
            d = spos[p0:p1] - pos[qi]; d2s.append((d * d).sum(1))
        d2 = np.sort(np.concatenate(d2s)); p1c[qi] = len(d2)
        rho2 = d2[K - 1] if len(d2) >= K else np.inf
        if not np.isfinite(rho2):
            rho2 = (4 * h) ** 2  # fallback (rare)
        rho = math.sqrt(rho2)
        # optional middle pass: ring R1+1 trimmed with rho, then re-tighten rho (costs one more drain)
        # pass 2: rows intersecting the disc
        ya = max(int(math.floor((y[qi] - rho - gy0) / h)), 0); yb = min(int(math.floor((y[qi] + rho - gy0) / h)), Gy - 1)
        tot = 0; runs = 0
        for yy in range(ya, yb + 1):
            y0r = gy0 + yy * h; y1r = y0r + h
            gy = 0.0 if y0r <= y[qi] <= y1r else min(abs(y[qi] - y0r), abs(y[qi] - y1r))
            rem = rho2 - gy * gy
            if rem < 0: continue
            xr = math.sqrt(rem)
            a = max(int(math.floor((x[qi] - xr - gx0) / h)), 0); b = min(int(math.floor((x[qi] + xr - gx0) / h)), Gx - 1)
            if a > b: continue
            inner = abs(yy - cy[qi]) <= R1
            if inner:
                # exclude [cx-R1, cx+R1]
                segs = [(a, min(b, cx[qi] - R1 - 1)), (max(a, cx[qi] + R1 + 1), b)]
            else:
                segs = [(a, b)]
            for (sa, sb) in segs:
                if sa > sb: continue
                c = cs[yy * Gx + sb + 1] - cs[yy * Gx + sa]
                if c > 0: tot += c; runs += 1
        p2c[qi] = tot; p2runs[qi] = runs
    if order_mode == "cell": qo = order
    slots = 0; waves = 0; s1 = 0; s2 = 0
    for w0 in range(0, n, 64):
        q = qo[w0:w0 + 64]
        a = p1c[q].max(); b = (p2c[q] + 2 * p2runs[q]).max()  # ~2 wasted slots per run (partial trips of 4)
        s1 += a; s2 += b; waves += 1
    return p1c.mean(), p2c.mean(), p2runs.mean(), s1 / waves, s2 / waves

for ct in (7.0, 5.0, 4.0, 3.0):
    for R1 in (1,) if ct > 3.5 else (1, 2):
        a, b, r, s1, s2 = run(0, ct, R1=R1)
        print(f"cell_target {ct} R1={R1}: pass1 cand {a:.0f}, pass2 cand {b:.0f} in {r:.1f} runs | wave slots: pass1 {s1:.0f} + pass2 {s2:.0f} = {s1+s2:.0f}")
