"""BASELINE configs[2] end to end (missing #4): one synthetic cloud through the reference's whole inference chain
— tile selection -> GridSampling -> node budget -> normalisations -> forward -> knn_interpolate(k=10) -> scatter_sum merge ->
softmax / argmax / entropy (``/root/reference/myria3d/predict.py:49-66``) — on the MI355X (``myria3d_amd.predict_cloud``)
against the same chain made of oracle pieces on the CPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cloud(side_m: float, density: float, seed: int):
    """Lidar-HD-shaped points over a side_m x side_m square with a projected-coordinate-style offset (kept small enough for fp32 to resolve centimetres), raw features (Intensity as
    counts, colours 0..255)."""
    rs = np.random.RandomState(seed)
    n = int(side_m * side_m * density)
    xy = rs.uniform(0, side_m, (n, 2))
    z0 = 2.0 * np.sin(2 * np.pi * xy[:, 0] / 50.0) + 1.5 * np.cos(2 * np.pi * xy[:, 1] / 37.0)
    u = rs.uniform(size=n)
    z = np.where(u < 0.5, z0 + rs.normal(0, 0.05, n), np.where(u < 0.85, z0 + rs.uniform(0, 15, n), z0 + rs.uniform(3, 9, n)))
    pos = np.concatenate([xy + np.array([8430.0, 65190.0]), (z + 200.0)[:, None]], 1).astype(np.float32)
    x = rs.uniform(0, 1, (n, 9)).astype(np.float32)
    x[:, 0] = rs.gamma(2.0, 300.0, n).astype(np.float32)
    x[:, 7] = rs.uniform(0, 255, n).astype(np.float32)
    return torch.from_numpy(pos), torch.from_numpy(x)


def _oracle_chain(ref, pos, x, tile, sub, overlap, batch_size, k, dec_seed):
    from oracle import prep_oracle as O
    from oracle.randla_oracle import fixed_decimation_indices, knn_interpolate

    samples = [(s, np.sort(i)) for s, i in O.split_cloud_into_samples(pos.numpy(), tile, sub, overlap)]
    acc = torch.zeros((pos.shape[0], 6), dtype=torch.float32)
    stored_idx = []
    for b0 in range(0, len(samples), batch_size):
        chunk = samples[b0:b0 + batch_size]
        rows = torch.from_numpy(np.concatenate([i for _, i in chunk])).long()
        sizes = [len(i) for _, i in chunk]
        ptr_full = [0] + list(np.cumsum(sizes))
        pos_copy, x_raw = pos[rows], x[rows]
        # GridSampling per sample (positions of the sub-sampled copy = voxel means, before Center)
        subs = [O.grid_sampling(pos_copy[s:e], x_raw[s:e], None, 0.25) for s, e in zip(ptr_full[:-1], ptr_full[1:])]
        pos_sampled = torch.cat([q[0] for q in subs])
        pn, xn, _, ptr = O.prepare_tiles(pos_copy, x_raw, None, ptr_full, 0.25, sub, 0, 7)
        assert ptr[-1] == pos_sampled.shape[0]
        assert all(300 <= b - a <= 40000 for a, b in zip(ptr[:-1], ptr[1:])), "test data must stay inside the node budget"
        batch = torch.repeat_interleave(torch.arange(len(chunk)), torch.tensor([b - a for a, b in zip(ptr[:-1], ptr[1:])]))
        with torch.no_grad():
            logits = ref(xn, pn, batch, torch.tensor(ptr), decimation_idx=fixed_decimation_indices(ptr, 4, seed=dec_seed))
            full = knn_interpolate(logits, pos_sampled, pos_copy, ptr, ptr_full, k=k)
        acc.index_add_(0, rows, full)
        stored_idx.append(rows)
    idx = torch.cat(stored_idx)
    probas = torch.softmax(acc[idx], dim=1)
    return acc, idx, probas, probas.argmax(1), torch.distributions.Categorical(probs=probas).entropy()


@pytest.mark.parametrize("overlap", [0, 10])
def test_predict_chain_end_to_end_matches_the_oracle_chain(overlap):
    from myria3d_amd import HipRandLANet, predict_cloud
    from oracle.randla_oracle import RandLANetOracle, fixed_decimation_indices
    from tests._util import fill_params_deterministic

    dev = torch.device("cuda:0")
    tile, sub = 150, 50  # 3 x 3 samples (4 x 4 overlapping ones) of ~7 500 raw points: 67 500 points in all
    pos, x = _cloud(tile, 3.0, seed=overlap)
    ref = RandLANetOracle(9, 6, return_logits=True)
    fill_params_deterministic(ref, 5)
    ref.eval()
    net = HipRandLANet(9, 6, return_logits=True)
    net.load_state_dict(ref.state_dict())
    net = net.to(dev).eval()
    acc, idx, probas, preds, entropy = _oracle_chain(ref, pos, x, tile, sub, overlap, batch_size=4, k=10, dec_seed=3)
    out = predict_cloud(net, pos.to(dev), x.to(dev), tile_width=tile, subtile_width=sub, subtile_overlap=overlap, batch_size=4,
                        decimation_idx_fn=lambda ptr: fixed_decimation_indices(ptr, 4, seed=3))
    assert torch.equal(out["idx_in_full_cloud"].cpu().long(), idx), "stored predictions: same points in the same order"
    err = (out["logits_full"].cpu() - acc).abs().max().item()
    print(f"[parity] predict chain overlap={overlap}: {idx.numel()} stored predictions over {pos.shape[0]} points, "
          f"max |merged logit - oracle| = {err:.3e} (|logit| max {acc.abs().max().item():.2f})")
    assert torch.allclose(out["logits_full"].cpu(), acc, rtol=2e-4, atol=2e-4 * max(1.0, acc.abs().max().item()))
    assert torch.allclose(out["probas"].cpu(), probas, rtol=1e-3, atol=2e-4)
    agree = (out["preds"].cpu() == preds).float().mean().item()
    assert agree >= 0.999, agree
    assert torch.allclose(out["entropy"].cpu(), entropy, rtol=1e-3, atol=1e-3)
    covered = torch.zeros(pos.shape[0], dtype=torch.bool)
    covered[idx] = True
    assert covered.all(), "every point of the cloud belongs to at least one sample"
    if overlap:
        assert idx.numel() > pos.shape[0]  # overlapping samples predict some points twice: their logits were summed


def test_predict_chain_sharded_over_two_ranks_merges_to_the_same_logits():
    """Samples of ONE cloud sharded over ranks (rank r takes samples r, r + W, ...): the rank-local accumulators add up to the
    single-rank merge (what the all-reduce of ``predict_cloud(world_size > 1)`` computes; emulated here on one GPU)."""
    from myria3d_amd import HipRandLANet, predict_cloud
    from oracle.randla_oracle import fixed_decimation_indices

    dev = torch.device("cuda:0")
    pos, x = _cloud(100, 3.0, seed=7)
    torch.manual_seed(1)
    net = HipRandLANet(9, 6, return_logits=True).to(dev).eval()
    kw = dict(tile_width=100, subtile_width=50, subtile_overlap=10, batch_size=1,
              decimation_idx_fn=lambda ptr: fixed_decimation_indices(ptr, 4, seed=2))
    whole = predict_cloud(net, pos.to(dev), x.to(dev), **kw)
    shards = []

    class _FakeDist:
        @staticmethod
        def all_reduce(t, group=None):
            shards.append(t.clone())
    import torch.distributed as dist
    real = dist.all_reduce
    dist.all_reduce = _FakeDist.all_reduce
    try:
        outs = [predict_cloud(net, pos.to(dev), x.to(dev), **kw, rank=r, world_size=2) for r in range(2)]
    finally:
        dist.all_reduce = real
    acc = shards[0] + shards[2]  # (logit accumulators of rank 0 and rank 1; shards[1], shards[3] are the hit counters)
    hit = shards[1] + shards[3]
    assert bool((hit > 0).all())
    assert torch.allclose(acc, whole["logits_full"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("batch_size", [1, 4])
def test_predict_chain_lookahead_of_the_position_only_work_changes_nothing(batch_size):
    """Round 6: ``predict_cloud`` builds the position-only tables of batch b + 1 (the interpolation's k-NN table, the net's
    grids / K-NN tables / decimation draw / decoder 1-NN tables through ``HipRandLANet.prefetch_geometry``) on side streams
    while batch b's feature kernels run.  Same kernels on the same inputs, the net drawing its own decimation either way:
    the merged logits are BIT-identical with and without, on layouts that change with every batch (9 samples of different
    sizes, the last batch shorter; no overlap between samples: a point predicted three times would be summed in the order its
    atomics arrive, lookahead or not)."""
    from myria3d_amd import HipRandLANet, predict_cloud

    dev = torch.device("cuda:0")
    pos, x = _cloud(150, 3.0, seed=11)
    torch.manual_seed(2)
    net = HipRandLANet(9, 6, return_logits=True).to(dev).eval()
    outs = []
    for look in (False, True, True):
        net.set_decimation_seed(77)
        outs.append(predict_cloud(net, pos.to(dev), x.to(dev), tile_width=150, subtile_width=50, subtile_overlap=0,
                                  batch_size=batch_size, lookahead=look))
    for o in outs[1:]:
        assert torch.equal(o["idx_in_full_cloud"], outs[0]["idx_in_full_cloud"])
        assert torch.equal(o["logits_full"], outs[0]["logits_full"])
        assert torch.equal(o["preds"], outs[0]["preds"])
    assert bool(torch.isfinite(outs[0]["logits_full"]).all())
    assert not net._look_queue, "every prefetched table set was consumed by its forward"
