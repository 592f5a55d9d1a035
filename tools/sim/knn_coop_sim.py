"""Ideal candidate counts of a wavefront-cooperative search: 64 consecutive queries (in some spatial order) scan every
point of the bounding box of their K-NN balls."""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myria3d_amd.synthetic import synthetic_tile
from scipy.spatial import cKDTree

def morton3(ix, iy, iz, bits=7):
    def spread(v):
        r = np.zeros_like(v)
        for b in range(bits):
            r |= ((v >> b) & 1) << (3 * b)
        return r
    return spread(ix) | (spread(iy) << 1) | (spread(iz) << 2)

def morton2(ix, iy, bits=8):
    def spread(v):
        r = np.zeros_like(v)
        for b in range(bits):
            r |= ((v >> b) & 1) << (2 * b)
        return r
    return spread(ix) | (spread(iy) << 1)

def box_counts(pos, order, rk, G=64, quant=None):
    n = len(pos)
    out = []
    for w0 in range(0, n, G):
        q = order[w0:w0 + G]
        lo = (pos[q] - rk[q, None]).min(0); hi = (pos[q] + rk[q, None]).max(0)
        if quant is not None:  # snap the box outward to cells of size quant
            lo = np.floor(lo / quant) * quant; hi = np.ceil(hi / quant) * quant
        m = np.all((pos >= lo) & (pos <= hi), axis=1)
        out.append(m.sum())
    return np.array(out)

K = 16
for tid in range(2):
    _, pos, _ = synthetic_tile(12800, tid)
    pos = pos.numpy().astype(np.float64)
    n = len(pos)
    tree = cKDTree(pos)
    dk, _ = tree.query(pos, k=K)
    rk = dk[:, -1]
    mn = pos.min(0); ext = (pos.max(0) - mn).max()
    for G in (64, 32, 16):
        for name, cell in (("morton3 h=.047", 0.047), ("morton3 h=.03", 0.03), ("morton3 h=.02", 0.02)):
            iq = ((pos - mn) / cell).astype(np.int64)
            code = morton3(iq[:, 0], iq[:, 1], iq[:, 2])
            order = np.argsort(code, kind="stable")
            c = box_counts(pos, order, rk, G)
            cq = box_counts(pos, order, rk, G, quant=0.047)
            print(f"tile {tid} G={G} {name}: ideal box candidates per group mean {c.mean():.0f} median {np.median(c):.0f} p90 {np.percentile(c,90):.0f} max {c.max()} | snapped to 0.047 cells: mean {cq.mean():.0f}")
        # 2-D cell order (current)
        h = 0.047
        cx = ((pos[:, 0] - mn[0]) / h).astype(int); cy = ((pos[:, 1] - mn[1]) / h).astype(int)
        order = np.lexsort((cx, cy))
        c = box_counts(pos, order, rk, G)
        print(f"tile {tid} G={G} row-major 2-D cells: mean {c.mean():.0f} median {np.median(c):.0f} p90 {np.percentile(c,90):.0f}")
        code = morton2(cx, cy)
        order = np.argsort(code, kind="stable")
        c = box_counts(pos, order, rk, G)
        print(f"tile {tid} G={G} morton2 cells: mean {c.mean():.0f} median {np.median(c):.0f} p90 {np.percentile(c,90):.0f}")
