"""CPU model (numpy float32, same expressions as ``csrc/knn.hip``) of the circular-ring walk of the deferred-insertion kNN
kernel: trimming a ring's runs to the disc of the current k-th distance must never drop a neighbour, whatever the rounding —
the walk's result equals the brute-force table of the oracle (ties by index).  Checks the ARGUMENT the kernel relies on (the
HIP code itself is compared with the oracle on the GPU: ``tests/test_gpu_ops.py``)."""
import math

import numpy as np
import pytest
import torch

from oracle.randla_oracle import knn_exact, synthetic_batch

f32 = np.float32


def _grid(pos, cell_target=7.0, gmax=64):
    n = len(pos)
    x, y = pos[:, 0], pos[:, 1]
    xmin, xmax, ymin, ymax = x.min(), x.max(), y.min(), y.max()
    wx, wy = f32(xmax - xmin), f32(ymax - ymin)
    wmax = max(wx, wy)
    if not wmax > 0:
        h = f32(1.0)
    else:
        area = f32(max(wx, f32(wmax * f32(1e-3))) * max(wy, f32(wmax * f32(1e-3))))
        h = f32(math.sqrt(f32(area * f32(cell_target) / f32(n))))
        h = max(h, f32(f32(wmax / f32(gmax)) * f32(1.0001)))
    Gx, Gy = min(gmax, int(f32(wx / h)) + 1), min(gmax, int(f32(wy / h)) + 1)
    amax = max(abs(xmin), abs(xmax), abs(ymin), abs(ymax))
    inv_h = f32(f32(1.0) / h)
    eps = f32(f32(2e-4) * h + f32(16.0) * f32(1.1920929e-7) * f32(amax))
    cx = np.clip((f32(x - xmin) * inv_h).astype(np.int64), 0, Gx - 1)
    cy = np.clip((f32(y - ymin) * inv_h).astype(np.int64), 0, Gy - 1)
    cell = cy * Gx + cx
    order = np.argsort(cell, kind="stable")
    cs = np.zeros(Gx * Gy + 1, np.int64)
    np.add.at(cs, cell + 1, 1)
    return dict(gx0=f32(xmin), gy0=f32(ymin), h=h, inv_h=inv_h, eps=eps, Gx=Gx, Gy=Gy, cs=np.cumsum(cs), order=order, cx=cx, cy=cy)


def _walk(pos, g, qi, K, trim=True):
    """the kernel's ring walk for query row qi; returns the sorted (d2, row) list and the number of candidates examined"""
    qx, qy, qz = pos[qi]
    cx, cy = g["cx"][qi], g["cy"][qi]
    best = []  # (d2, row)
    kth = f32(np.inf)
    seen = 0
    R = 0
    while True:
        k2 = f32(kth * f32(1.0000153)) if trim else f32(np.inf)
        new = []
        for dy in range(-R, R + 1):
            yy = cy + dy
            if yy < 0 or yy >= g["Gy"]:
                continue
            rem = k2
            if trim and dy != 0:
                edge = f32(f32(g["gy0"] + f32(f32(yy) * g["h"])) - qy) if dy > 0 else f32(qy - f32(g["gy0"] + f32(f32(yy + 1) * g["h"])))
                gap = max(f32(edge - g["eps"]), f32(0))
                rem = f32(k2 - f32(gap * gap))
                if rem < 0:
                    continue
            if abs(dy) == R:
                segs = [(max(cx - R, 0), min(cx + R, g["Gx"] - 1), True)]
            else:
                segs = [(cx - R, cx - R, False), (cx + R, cx + R, False)]
            for sg, (xa, xb, edge_row) in enumerate(segs):
                if edge_row:
                    if trim and rem < f32(3.0e38):
                        xr = f32(f32(np.sqrt(rem)) * f32(1.000001) + g["eps"])
                        fa = min(max(f32(f32(f32(qx - xr) - g["gx0"]) * g["inv_h"]), f32(0)), f32(65535))
                        fb = min(max(f32(f32(f32(qx + xr) - g["gx0"]) * g["inv_h"]), f32(0)), f32(65535))
                        xa, xb = max(xa, int(fa)), min(xb, int(fb))
                else:
                    if xa < 0 or xa >= g["Gx"]:
                        continue
                    if trim:
                        ex = f32(qx - f32(g["gx0"] + f32(f32(xa + 1) * g["h"]))) if sg == 0 else f32(f32(g["gx0"] + f32(f32(xa) * g["h"])) - qx)
                        gx = max(f32(ex - g["eps"]), f32(0))
                        if f32(gx * gx) > rem:
                            continue
                if xa > xb:
                    continue
                p0, p1 = g["cs"][yy * g["Gx"] + xa], g["cs"][yy * g["Gx"] + xb + 1]
                rows = g["order"][p0:p1]
                seen += len(rows)
                d = pos[rows] - pos[qi]
                d2 = f32(f32(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
                for v, r in zip(d2, rows):
                    if not v > kth:  # admission test of the kernel (kth = value at the start of the ring)
                        new.append((v, int(r)))
        best = sorted(best + new)[:K]
        kth = best[K - 1][0] if len(best) >= K else f32(np.inf)
        covers = cx - R <= 0 and cx + R >= g["Gx"] - 1 and cy - R <= 0 and cy + R >= g["Gy"] - 1
        if covers:
            break
        b = f32(3.4e38)
        if cx - R > 0: b = min(b, f32(qx - f32(g["gx0"] + f32(f32(cx - R) * g["h"]))))
        if cx + R < g["Gx"] - 1: b = min(b, f32(f32(g["gx0"] + f32(f32(cx + R + 1) * g["h"])) - qx))
        if cy - R > 0: b = min(b, f32(qy - f32(g["gy0"] + f32(f32(cy - R) * g["h"]))))
        if cy + R < g["Gy"] - 1: b = min(b, f32(f32(g["gy0"] + f32(f32(cy + R + 1) * g["h"])) - qy))
        b = max(f32(b - g["eps"]), f32(0))
        if kth <= f32(b * b):
            break
        R += 1
    return best, seen


def _cases():
    rng = np.random.RandomState(0)
    yield "uniform", rng.rand(1500, 3).astype(f32), 16
    _, pos, _, _, _ = synthetic_batch([1800])
    yield "lidar", pos.numpy(), 16
    yield "lidar k10", pos.numpy()[:900], 10
    dense = np.concatenate([rng.rand(600, 3) * 0.01 + 0.5, rng.rand(700, 3)]).astype(f32)
    yield "dense cluster", dense, 16
    big = (rng.rand(900, 3) * np.array([50.0, 50.0, 20.0]) + np.array([843000.0, 6519000.0, 200.0])).astype(f32)
    yield "lambert offsets", big, 16
    lat = np.stack(np.meshgrid(np.arange(30), np.arange(30), [0.0]), -1).reshape(-1, 3).astype(f32) * f32(0.125)
    yield "lattice (ties)", lat, 16
    yield "fewer than k", rng.rand(9, 3).astype(f32), 16


@pytest.mark.parametrize("name,pos,K", list(_cases()), ids=[c[0] for c in _cases()])
def test_circular_ring_walk_equals_brute_force(name, pos, K):
    g = _grid(pos)
    n = len(pos)
    ref_idx, ref_d2 = knn_exact(torch.from_numpy(pos), [0, n], torch.from_numpy(pos), [0, n], K)
    tot_trim = tot_sq = 0
    step = max(1, n // 400)  # a few hundred queries per case keep the CPU suite short
    for qi in range(0, n, step):
        best, seen = _walk(pos, g, qi, K, trim=True)
        rows = [r for _, r in best] + [-1] * (K - len(best))
        assert rows == ref_idx[qi].tolist(), (name, qi)
        assert [float(v) for v, _ in best] == [float(v) for v in ref_d2[qi][: len(best)]]
        tot_trim += seen
        tot_sq += _walk(pos, g, qi, K, trim=False)[1]
    assert tot_trim <= tot_sq
    print(f"[knn model] {name}: candidates per query {tot_trim / math.ceil(n / step):.0f} (square rings {tot_sq / math.ceil(n / step):.0f})")
