"""Per-lane walk in a 3-D grid: candidates per query for cubic shells of cells (anisotropic hz allowed)."""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myria3d_amd.synthetic import synthetic_tile
from scipy.spatial import cKDTree
from knn_coop_sim import morton3

def profile3d(pos, K, h, hz):
    n = len(pos)
    mn = pos.min(0)
    hv = np.array([h, h, hz])
    G = np.floor((pos.max(0) - mn) / hv).astype(int) + 1
    c = np.minimum(((pos - mn) / hv).astype(int), G - 1)
    cnt = np.zeros(G[::-1], int)  # z, y, x
    np.add.at(cnt, (c[:, 2], c[:, 1], c[:, 0]), 1)
    ps = np.zeros((G[2] + 1, G[1] + 1, G[0] + 1), int)
    ps[1:, 1:, 1:] = cnt.cumsum(0).cumsum(1).cumsum(2)
    def block(R):
        x0, x1 = np.maximum(c[:, 0] - R, 0), np.minimum(c[:, 0] + R, G[0] - 1) + 1
        y0, y1 = np.maximum(c[:, 1] - R, 0), np.minimum(c[:, 1] + R, G[1] - 1) + 1
        z0, z1 = np.maximum(c[:, 2] - R, 0), np.minimum(c[:, 2] + R, G[2] - 1) + 1
        return (ps[z1, y1, x1] - ps[z0, y1, x1] - ps[z1, y0, x1] - ps[z1, y1, x0] + ps[z0, y0, x1] + ps[z0, y1, x0] + ps[z1, y0, x0] - ps[z0, y0, x0])
    dk, _ = cKDTree(pos).query(pos, k=K)
    rk = dk[:, -1]
    Rn = np.zeros(n, int); done = np.zeros(n, bool)
    for R in range(0, G.max() + 1):
        b = np.full(n, 3.4e38)
        for ax in range(3):
            m = c[:, ax] - R > 0; b[m] = np.minimum(b[m], pos[m, ax] - (mn[ax] + (c[m, ax] - R) * hv[ax]))
            m = c[:, ax] + R < G[ax] - 1; b[m] = np.minimum(b[m], (mn[ax] + (c[m, ax] + R + 1) * hv[ax]) - pos[m, ax])
        ok = (rk <= b) & (block(R) >= K)
        Rn[ok & ~done] = R; done |= ok
        if done.all(): break
    cum = np.stack([block(R) for R in range(Rn.max() + 1)], 1)
    return G, c, Rn, cum

def wave_cost(order, Rn, cum, W=64):
    n = len(order); tot = 0
    for w0 in range(0, n, W):
        q = order[w0:w0 + W]
        for R in range(Rn[q].max() + 1):
            act = q[Rn[q] >= R]
            ring = cum[act, R] - (cum[act, R - 1] if R > 0 else 0)
            tot += ring.max()
    return tot / math.ceil(n / W)

K = 16
_, pos, _ = synthetic_tile(12800, 0)
pos = pos.numpy().astype(np.float64)
n = len(pos)
for h, hz in ((0.047, 0.047), (0.047, 0.094), (0.047, 0.14), (0.035, 0.07), (0.03, 0.06), (0.03, 0.12), (0.025, 0.1), (0.06, 0.06), (0.06, 0.12)):
    G, c, Rn, cum = profile3d(pos, K, h, hz)
    avg = cum[np.arange(n), Rn].mean()
    cell = (c[:, 2] * G[1] + c[:, 1]) * G[0] + c[:, 0]
    o_cell = np.argsort(cell, kind="stable")
    o_m = np.argsort(morton3(c[:, 0], c[:, 1], c[:, 2]), kind="stable")
    o_ring = np.lexsort((cell, Rn))
    runs = np.array([(2 * R + 1) ** 2 for R in range(12)])  # rows of cells visited through ring R (z-major rows)
    print(f"h={h} hz={hz} G={G} cells={G.prod()} cand/query {avg:.0f} rings mean {Rn.mean():.2f} hist {np.round(np.bincount(Rn, minlength=6)[:6]/n,2)} "
          f"rows/query {runs[Rn].mean():.0f} | wave steps: zyx-cell order {wave_cost(o_cell, Rn, cum):.0f} morton {wave_cost(o_m, Rn, cum):.0f} by-ring {wave_cost(o_ring, Rn, cum):.0f}")
