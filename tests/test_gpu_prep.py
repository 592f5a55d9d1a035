"""Device-side data preparation against oracle/prep_oracle.py: GridSampling, node budget,
normalisations.  Runs on the MI355X through the C ABI."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _lidar_like(rs, n, extent=(50.0, 50.0, 12.0), origin=(0.0, 0.0, 0.0)):
    pos = rs.uniform(0, 1, (n, 3)).astype(np.float32) * np.asarray(extent, np.float32) + np.asarray(origin, np.float32)
    x = rs.uniform(0, 1, (n, 9)).astype(np.float32)
    x[:, 0] = rs.gamma(2.0, 300.0, n).astype(np.float32)          # raw Intensity (log-standardised later)
    x[:, 7] = rs.uniform(0, 255, n).astype(np.float32)            # rgb_avg
    y = rs.randint(0, 6, n).astype(np.int64)
    return torch.from_numpy(pos), torch.from_numpy(x), torch.from_numpy(y)


def _batch(rs, sizes, **kw):
    P, X, Y = zip(*[_lidar_like(rs, n, origin=(1000.0 * i, -300.0 * i, 40.0 * i), **kw) for i, n in enumerate(sizes)])
    ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64)
    return torch.cat(P), torch.cat(X), torch.cat(Y), ptr


# ([90000, 45000]: 256 x 66 radix-histogram entries per pass — the multi-workgroup prefix sum of round 5, exscan_launch)
@pytest.mark.parametrize("sizes,size", [([30000, 12000], 0.25), ([5000], 1.0), ([1, 700, 2], 0.25), ([4000, 4000, 4000], 5.0),
                                        ([90000, 45000], 0.25)])
def test_grid_sampling_matches_pyg_semantics(device, sizes, size):
    from myria3d_amd import transforms as T
    from oracle import prep_oracle as O

    rs = np.random.RandomState(len(sizes) * 7 + int(size * 4))
    pos, x, y, ptr = _batch(rs, sizes)
    pos[:50] = pos[0]                 # 50 coincident points: one crowded voxel (majority vote over > 1 label)
    gp, gx, gy, gptr = T.grid_sampling(pos.to(device), x.to(device), y.to(device), ptr.to(device), size)
    ref = [O.grid_sampling(pos[s:e], x[s:e], y[s:e], size) for s, e in zip(ptr[:-1].tolist(), ptr[1:].tolist())]
    rptr = np.cumsum([0] + [r[0].shape[0] for r in ref])
    assert gptr.cpu().tolist() == rptr.tolist(), "number of occupied voxels per tile"
    rp, rx, ry = (torch.cat([r[k] for r in ref]) for k in range(3))
    assert gp.shape == rp.shape and gx.shape == rx.shape
    # sums run in original point order on both sides: means agree to the last bit or two
    assert torch.allclose(gp.cpu(), rp, rtol=1e-6, atol=1e-6), (gp.cpu() - rp).abs().max()
    assert torch.allclose(gx.cpu(), rx, rtol=1e-6, atol=1e-6), (gx.cpu() - rx).abs().max()
    print(f"[parity] grid_sampling sizes={sizes} voxels={rp.shape[0]} bit-equal pos={torch.equal(gp.cpu(), rp)} "
          f"x={torch.equal(gx.cpu(), rx)}")
    assert torch.equal(gy.cpu(), ry), "majority label, first maximum on ties"
    # pos only (x and y absent)
    gp2, gx2, gy2, gptr2 = T.grid_sampling(pos.to(device), None, None, ptr.to(device), size)
    assert gx2 is None and gy2 is None and torch.equal(gp2, gp) and torch.equal(gptr2, gptr)


def test_grid_sampling_properties_at_full_tile_size(device):
    """One 1 km2-style batch: 16 raw tiles of ~80 000 points; properties that need no O(N) reference loop."""
    from myria3d_amd import transforms as T

    rs = np.random.RandomState(3)
    sizes = [80000] * 16
    pos, x, y, ptr = _batch(rs, sizes, extent=(50.0, 50.0, 6.0))
    gp, gx, gy, gptr = T.grid_sampling(pos.to(device), x.to(device), y.to(device), ptr.to(device), 0.25)
    m = gp.shape[0]
    assert int(gptr[-1]) == m and bool((gptr[1:] > gptr[:-1]).all())
    # sampling the voxel means again (new bounding box, shifted cells) can only merge points, never create any
    gp2, _, _, gptr2 = T.grid_sampling(gp, None, None, gptr, 0.25)
    assert 0.4 * m <= gp2.shape[0] <= m
    # a coarser grid gives fewer voxels; a grid finer than any point spacing keeps (almost) every point
    assert T.grid_sampling(pos.to(device), None, None, ptr.to(device), 1.0)[0].shape[0] < m
    # every output point lies inside its tile's bounding box, near the tile centroid; labels / features stay in range
    for b in range(16):
        s, e = int(gptr[b]), int(gptr[b + 1])
        pts = pos[ptr[b]:ptr[b + 1]]
        lo, hi = pts.min(0).values, pts.max(0).values
        q = gp[s:e].cpu()
        assert bool((q >= lo - 1e-3).all()) and bool((q <= hi + 1e-3).all())
        assert float((q.mean(0) - pts.mean(0)).abs().max()) < 1.0
        # occupied voxels: between the 2-D footprint (200 x 200 columns) and the number of points
        assert 200 * 200 * 0.8 <= e - s <= 80000
    assert float(gx.min()) >= float(x.min()) - 1e-3 and float(gx.max()) <= float(x.max()) + 1e-3
    assert int(gy.min()) >= 0 and int(gy.max()) <= 5


def test_node_budget_follows_the_reference_sampling_scheme(device):
    from myria3d_amd import transforms as T

    rs = np.random.RandomState(11)
    sizes = [120, 5000, 900, 45000, 1]
    pos, x, y, ptr = _batch(rs, sizes)
    p, xx, yy, optr, idx = T.node_budget(pos.to(device), x.to(device), y.to(device), ptr.to(device), minimum=300,
                                         maximum=40000, seed=5)
    out = (optr[1:] - optr[:-1]).cpu().tolist()
    assert out == [300, 5000, 900, 40000, 300]
    idx = idx.cpu().long()
    assert torch.equal(p.cpu(), pos[idx]) and torch.equal(xx.cpu(), x[idx]) and torch.equal(yy.cpu(), y[idx])
    o = optr.cpu().tolist()
    # tile 0 (120 -> 300): ceil(300 / 120) = 3 permutations of the tile, concatenated, cut at 300
    t0 = idx[o[0]:o[1]]
    assert sorted(t0[:120].tolist()) == list(range(120)) and sorted(t0[120:240].tolist()) == list(range(120))
    assert len(set(t0[240:].tolist())) == 60 and t0[:120].tolist() != t0[120:240].tolist()
    # untouched tiles keep their order
    assert idx[o[1]:o[2]].tolist() == list(range(120, 5120)) and idx[o[2]:o[3]].tolist() == list(range(5120, 6020))
    # tile 3 (45 000 -> 40 000): distinct rows of that tile, not the first 40 000, roughly uniform
    t3 = idx[o[3]:o[4]] - 6020
    assert len(set(t3.tolist())) == 40000 and int(t3.min()) >= 0 and int(t3.max()) < 45000
    assert abs(float(t3.float().mean()) / 45000 - 0.5) < 0.01 and t3.tolist() != sorted(t3.tolist())
    # single-point tile repeated
    assert idx[o[4]:o[5]].tolist() == [6020 + 45000] * 300
    # another seed, another draw
    _, _, _, _, idx2 = T.node_budget(pos.to(device), None, None, ptr.to(device), minimum=300, maximum=40000, seed=6)
    assert not torch.equal(idx2.cpu().long(), idx)


def test_normalisations_match_the_reference_arithmetic(device):
    from myria3d_amd import transforms as T
    from oracle import prep_oracle as O

    rs = np.random.RandomState(2)
    sizes = [9000, 1, 2500]
    pos, x, y, ptr = _batch(rs, sizes)
    gp, gx = T.normalize_tiles(pos.to(device), x.to(device), ptr.to(device), center=True, nullify_z=True,
                               subtile_width=50, intensity_col=0, rgb_col=7)
    for b in range(3):
        s, e = int(ptr[b]), int(ptr[b + 1])
        rp = O.normalize_pos(O.nullify_lowest_z(O.center(pos[s:e])), 50)
        rx = O.standardize_rgb_and_intensity(x[s:e], 0, 7)
        # fp32 coordinates of ~1e3 m: the tile mean carries a few 1e-4 m of rounding in the reference's fp32 sum
        assert torch.allclose(gp[s:e].cpu(), rp, rtol=0, atol=1e-4), (gp[s:e].cpu() - rp).abs().max()
        assert torch.allclose(gx[s:e].cpu(), rx, rtol=1e-4, atol=1e-4), (gx[s:e].cpu() - rx).abs().max()
    assert float(gp[:, 2].min()) == 0.0
    # untouched columns are bit-identical, inputs are not modified
    keep = [c for c in range(9) if c not in (0, 7)]
    assert torch.equal(gx[:, keep].cpu(), x[:, keep])


def test_transform_objects_compose_like_the_reference_pipeline(device):
    """points_budget.yaml + normalizations/default.yaml as objects on a collated batch == oracle per tile."""
    from myria3d_amd import transforms as T
    from oracle import prep_oracle as O

    rs = np.random.RandomState(4)
    sizes = [20000, 15000]
    pos, x, y, ptr = _batch(rs, sizes)
    data = types.SimpleNamespace(pos=pos.to(device), x=x.to(device), y=y.to(device), ptr=ptr.to(device),
                                 batch=torch.repeat_interleave(torch.arange(2), torch.tensor(sizes)).to(device),
                                 x_features_names=["Intensity", "a", "b", "c", "d", "e", "f", "rgb_avg", "ndvi"])
    chain = [T.GridSampling(0.25), T.MinimumNumNodes(300), T.MaximumNumNodes(40000), T.Center(), T.NullifyLowestZ(),
             T.NormalizePos(subtile_width=50), T.StandardizeRGBAndIntensity()]
    for t in chain:
        data = t(data)
    rp, rx, ry, rptr = O.prepare_tiles(pos, x, y, ptr.tolist(), 0.25, 50, 0, 7)
    assert data.ptr.cpu().tolist() == rptr
    assert data.batch.shape[0] == data.pos.shape[0] == rptr[-1]
    assert torch.allclose(data.pos.cpu(), rp, rtol=0, atol=1e-4)
    assert torch.allclose(data.x.cpu(), rx, rtol=1e-4, atol=1e-4)
    assert torch.equal(data.y.cpu(), ry)
    # the prepared batch feeds the network unchanged
    import myria3d_amd
    net = myria3d_amd.HipRandLANet(9, 6, return_logits=True).to(device).eval()
    with torch.no_grad():
        logits = net(data.x, data.pos, data.batch, data.ptr)
    assert logits.shape == (rptr[-1], 6) and bool(torch.isfinite(logits).all())


def test_golden_fixture_prep_and_merge(device):
    """Committed vectors (tests/golden/prep_small.npz, generated by the CPU oracles): the kernels against the files,
    without running the oracle."""
    import os

    import myria3d_amd
    from myria3d_amd import transforms as T

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prep_small.npz"))
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    p, xx, yy, ptr = T.grid_sampling(t("pos").to(device), t("x").to(device), t("y").to(device), t("ptr").to(device), 0.25)
    assert ptr.cpu().tolist() == g["prep_ptr"].tolist() and torch.equal(yy.cpu(), t("prep_y"))
    p, xx = T.normalize_tiles(p, xx, ptr, center=True, nullify_z=True, subtile_width=50, intensity_col=0, rgb_col=7)
    assert torch.allclose(p.cpu(), t("prep_pos"), rtol=0, atol=1e-4)
    assert torch.allclose(xx.cpu(), t("prep_x"), rtol=1e-4, atol=1e-4)
    itp = myria3d_amd.DeviceInterpolator()
    for i in range(3):
        itp.store_predictions(t(f"logits{i}").to(device), [g[f"idx{i}"]])
    out = itp.reduce_predictions(int(g["nb_points"]))
    assert torch.equal(out["idx_in_full_cloud"].cpu(), t("cat_idx"))
    assert torch.allclose(out["probas"].cpu(), t("probas"), rtol=1e-6, atol=1e-5)
    assert torch.allclose(out["entropy"].cpu(), t("entropy"), rtol=2e-6, atol=1e-5)
    top2 = t("rows").topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(out["preds"].cpu()[clear], t("preds")[clear])
