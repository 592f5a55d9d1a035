"""Tiling of a cloud into square samples: oracle (scipy cKDTree, the reference's own call) vs brute
force on the CPU; the HIP kernel vs the oracle on the GPU (bit-exact membership, incl. points exactly on sample borders)."""
import numpy as np
import pytest
import torch

from oracle import prep_oracle as O


def _cloud(n, width, seed, lattice=False):
    rs = np.random.RandomState(seed)
    pos = rs.uniform(0, width, (n, 3)).astype(np.float32)
    if lattice:  # a third of the points sit exactly on multiples of 12.5 m: sample borders / centres
        k = n // 3
        pos[:k, :2] = (rs.randint(0, int(width / 12.5) + 1, (k, 2)) * 12.5).astype(np.float32)
    pos[:, :2] += np.float32(651234.5)  # Lambert-93-sized offsets: float32 spacing 1/16 m, like real LAS coordinates
    return pos


def _brute(pos, tile_width, subtile_width, overlap):
    xy = (pos[:, :2] - pos[:, :2].min(axis=0)).astype(np.float64)
    r = subtile_width // 2
    out = {}
    for s, c in enumerate(O.get_mosaic_of_centers(tile_width, subtile_width, overlap)):
        m = (np.abs(xy - c) <= r).all(axis=1)
        if m.any():
            out[s] = np.nonzero(m)[0]
    return out


@pytest.mark.parametrize("tile,sub,overlap", [(1000, 50, 0), (1000, 50, 25), (100, 50, 0), (110, 50, 10), (50, 50, 0)])
def test_oracle_tiling_equals_brute_force(tile, sub, overlap):
    pos = _cloud(6000, tile, seed=tile + overlap, lattice=True)
    got = {s: np.sort(i) for s, i in O.split_cloud_into_samples(pos, tile, sub, overlap)}
    ref = _brute(pos, tile, sub, overlap)
    assert sorted(got) == sorted(ref)
    for s in ref:
        assert np.array_equal(got[s], ref[s]), s


def test_mosaic_of_centers_contract():
    from myria3d_amd.tiling import get_mosaic_of_centers

    for args in ((1000, 50, 0), (1000, 50, 25), (110, 50, 10)):
        a, b = get_mosaic_of_centers(*args), O.get_mosaic_of_centers(*args)
        assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
    c = get_mosaic_of_centers(1000, 50, 0)
    assert len(c) == 400 and np.array_equal(c[0], [25.0, 25.0]) and np.array_equal(c[1], [25.0, 75.0])  # x-major
    # (the reference's own test, tests/myria3d/pctl/dataset/test_utils.py:7-15: centres stay inside the tile)
    for tw, sw, ov in ((1000, 50, 0), (1000, 50, 25)):
        m = np.stack(get_mosaic_of_centers(tw, sw, ov))
        assert m.min() - sw / 2 == 0 and m.max() + sw / 2 <= tw
    with pytest.raises(ValueError):
        get_mosaic_of_centers(1000, 50, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,tile,sub,overlap,lattice", [(200000, 1000, 50, 0, False), (60000, 1000, 50, 25, True),
                                                         (5000, 100, 50, 0, True), (3000, 110, 50, 10, True),
                                                         (1, 1000, 50, 0, False), (70, 50, 50, 0, True)])
def test_tile_select_kernel_matches_the_oracle(device, n, tile, sub, overlap, lattice):
    from myria3d_amd import tiling

    pos = _cloud(n, tile, seed=n + overlap, lattice=lattice)
    ref = {s: np.sort(i) for s, i in O.split_cloud_into_samples(pos, tile, sub, overlap)}
    sample_ptr, idx, centers = tiling.tile_select(torch.from_numpy(pos).to(device), tile, sub, overlap)
    sp, ix = sample_ptr.cpu().numpy(), idx.cpu().numpy()
    mosaic = O.get_mosaic_of_centers(tile, sub, overlap)
    assert sp.shape[0] == len(mosaic) + 1 and sp[0] == 0 and sp[-1] == ix.shape[0]
    assert np.array_equal(centers, np.stack(mosaic))
    nonempty = [s for s in range(len(mosaic)) if sp[s + 1] > sp[s]]
    assert nonempty == sorted(ref)
    for s in nonempty:
        assert np.array_equal(ix[sp[s]:sp[s + 1]], ref[s]), s  # same SET, ascending
    # the generator mirrors the reference's loop: non-empty samples in mosaic order
    gen = [g.cpu().numpy() for g in tiling.split_cloud_into_samples(torch.from_numpy(pos).to(device), tile, sub, overlap)]
    assert len(gen) == len(ref) and all(np.array_equal(g, ref[s]) for g, s in zip(gen, sorted(ref)))


@pytest.mark.gpu
def test_forward_like_model_branches(device):
    """Model.forward (model.py:67-103) around HipRandLANet: train / no-copies -> (y, logits) on the sub-sampled points;
    eval with copies -> logits interpolated (k = 10, on the device) onto every original point + transformed_y_copy."""
    import myria3d_amd
    from myria3d_amd import HipRandLANet, collate_tiles, forward_like_model
    from oracle.randla_oracle import knn_interpolate as ref_interp, synthetic_tile

    torch.manual_seed(0)
    tiles = []
    for t, (n_full, n_sub) in enumerate(((3000, 1200), (2500, 900))):
        x, pos, y = synthetic_tile(n_full, t)
        sel = torch.randperm(n_full, generator=torch.Generator().manual_seed(t))[:n_sub]
        tiles.append({"x": x[sel], "pos": pos[sel], "y": y[sel], "idx_in_original_cloud": np.arange(n_full) + 10 * t,
                      "copies": {"pos_sampled_copy": pos[sel].clone(), "pos_copy": pos.clone(),
                                 "transformed_y_copy": y.clone()}})
    batch = collate_tiles(tiles).to(device)
    net = HipRandLANet(9, 6, return_logits=True).to(device)
    net.train()
    targets, logits = forward_like_model(net, batch)
    assert logits.shape == (2100, 6) and torch.equal(targets, batch.y)
    net.eval()
    with torch.no_grad():
        net.set_decimation_seed(5)
        targets, dense = forward_like_model(net, batch, interpolation_k=10)
        net.set_decimation_seed(5)
        sub = net(batch.x, batch.pos, batch.batch, batch.ptr)
    assert dense.shape == (5500, 6) and torch.equal(targets, batch.copies["transformed_y_copy"])
    want = ref_interp(sub.cpu(), batch.copies["pos_sampled_copy"].cpu(), batch.copies["pos_copy"].cpu(), [0, 1200, 2100],
                      [0, 3000, 5500], k=10)
    assert torch.allclose(dense.cpu(), want, rtol=1e-4, atol=1e-5)
    nocopies = myria3d_amd.SimpleBatch(x=batch.x, pos=batch.pos, batch=batch.batch, ptr=batch.ptr, y=batch.y)
    with torch.no_grad():
        t2, l2 = forward_like_model(net, nocopies)
    assert l2.shape == (2100, 6) and torch.equal(t2, batch.y)
