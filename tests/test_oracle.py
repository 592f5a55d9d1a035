"""CPU: the oracle itself — pinned by the reference's shape cases, independent cross-checks of every restated
third-party semantic, and the committed golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import randla_oracle as O
from tests._util import fill_params_deterministic, rand_batch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "randla_small.npz")


@pytest.mark.parametrize("num_nodes,knn", [([50, 50], "exact"), ([1250, 1000], "exact"), ([12500, 10000], "kdtree")])
def test_reference_shape_cases(num_nodes, knn):
    """tests/myria3d/models/modules/test_randla_nets.py:8-40 restated: train mode, random decimation, shape only."""
    torch.manual_seed(0)
    net = O.RandLANetOracle(9, 6, decimation=4, num_neighbors=16, knn=knn)
    x, pos, batch, ptr = rand_batch(num_nodes)
    out = net(x, pos, batch, ptr)
    assert out.shape == torch.Size([sum(num_nodes), 6])
    assert torch.allclose(out.exp().sum(1), torch.ones(sum(num_nodes)), atol=1e-4)  # log_softmax by default


def test_state_dict_keys_follow_pyg_naming():
    sd = O.RandLANetOracle(9, 7).state_dict()
    for key, shape in {
        "fc0.weight": (32, 9), "block1.mlp1.lins.0.weight": (4, 32), "block1.shortcut.norms.0.module.running_mean": (32,),
        "block1.lfa1.mlp_encoder.lins.0.weight": (4, 10), "block1.lfa1.mlp_attention.lins.0.weight": (8, 8),
        "block4.lfa2.mlp_post_attention.norms.0.module.num_batches_tracked": (), "mlp_summit.lins.0.weight": (512, 512),
        "fp4.nn.lins.0.weight": (256, 768), "fp1.nn.lins.0.weight": (32, 64), "mlp_classif.lins.1.weight": (32, 64),
        "mlp_classif.norms.1.module.weight": (32,), "fc_classif.weight": (7, 32),
    }.items():
        assert tuple(sd[key].shape) == shape, key
    assert "block1.lfa1.mlp_attention.lins.0.bias" not in sd  # bias=False, norm=None
    assert sum(p.numel() for p in O.RandLANetOracle(9, 6).parameters()) == 1113686
    assert sum(p.numel() for p in O.RandLANetOracle(9, 7).parameters()) == 1113719


def test_knn_exact_vs_kdtree_and_conventions():
    x, pos, batch, ptr = rand_batch([700, 300, 5], seed=4)
    p = ptr.tolist()
    ie, de = O.knn_exact(pos, p, pos, p, 16)
    ik, dk = O.knn_kdtree(pos, p, pos, p, 16)
    assert torch.allclose(de[:1000], dk[:1000], rtol=1e-5, atol=1e-9)  # same distances (indices may differ at ties)
    assert (ie[:1000] == ik[:1000]).float().mean() > 0.999
    assert bool((ie[:, 0] == torch.arange(1005)).all())  # loop=True: self is the first neighbour
    assert bool((de[:, 1:] >= de[:, :-1]).all())
    assert bool((ie[1000:, 5:] == -1).all()) and bool((ie[1000:, :5] >= 1000).all())  # K_eff = min(K, n)
    ei = O.dense_to_edge_index(ie)
    assert ei.shape == (2, 1000 * 16 + 25) and bool((ei[1][:16] == 0).all())
    # large-n path (top-k + tie repair) equals the full stable sort
    pos_big = torch.cat([pos[:700].repeat(7, 1), pos[:200]])  # 5100 points with 7-fold duplicates
    a, _ = O.knn_exact(pos_big, [0, 5100], pos_big[:64], [0, 64], 16)
    d = ((pos_big[None, :, 0] - pos_big[:64, None, 0]) ** 2 + (pos_big[None, :, 1] - pos_big[:64, None, 1]) ** 2) + \
        (pos_big[None, :, 2] - pos_big[:64, None, 2]) ** 2
    assert torch.equal(a, torch.sort(d, dim=1, stable=True).indices[:, :16])


def test_segment_softmax_and_scatter_match_dense_torch():
    rs = np.random.RandomState(0)
    src = torch.from_numpy(rs.randn(40 * 16, 8).astype(np.float32))
    index = torch.arange(40).repeat_interleave(16)
    got = O.segment_softmax(src, index, 40)
    ref = torch.softmax(src.view(40, 16, 8), dim=1).view(-1, 8)
    assert torch.allclose(got, ref, atol=1e-6)
    assert torch.allclose(O.scatter_sum(src, index, 40), src.view(40, 16, 8).sum(1), atol=1e-5)


def test_shared_mlp_matches_stock_torch_modules():
    mlp = O.SharedMLP([10, 16, 8], dropout=[0.0, 0.5])
    fill_params_deterministic(mlp, 3)
    x = torch.randn(64, 10)
    mlp.train()
    y = mlp(x, dropout_masks=[None, torch.ones(64, 8)])
    h = x
    for lin, norm in zip(mlp.lins, mlp.norms):
        z = torch.nn.functional.linear(h, lin.weight, lin.bias)
        zn = (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + 1e-6) * norm.module.weight + norm.module.bias
        h = torch.nn.functional.leaky_relu(zn, 0.2)
    assert torch.allclose(y, h / 0.5, atol=1e-5)  # injected all-ones keep mask, p=0.5 scaling
    assert int(mlp.norms[0].module.num_batches_tracked) == 1
    assert mlp.norms[0].module.momentum == 0.01 and mlp.norms[0].module.eps == 1e-6


def test_knn_interpolate_semantics():
    rs = np.random.RandomState(1)
    pos_x = torch.from_numpy(rs.rand(50, 3).astype(np.float32))
    pos_y = torch.from_numpy(rs.rand(120, 3).astype(np.float32))
    x = torch.from_numpy(rs.randn(50, 4).astype(np.float32))
    y1 = O.knn_interpolate(x, pos_x, pos_y, [0, 50], [0, 120], k=1)
    nn = ((pos_y[:, None] - pos_x[None]) ** 2).sum(-1).argmin(1)
    assert torch.allclose(y1, x[nn], rtol=1e-6, atol=1e-6)  # k=1: x*w/w == x up to 1 ulp
    y3 = O.knn_interpolate(x, pos_x, pos_y, [0, 50], [0, 120], k=3)
    d2 = ((pos_y[:, None] - pos_x[None]) ** 2).sum(-1)
    dk, ik = d2.topk(3, largest=False)
    w = 1 / dk.clamp(min=1e-16)
    assert torch.allclose(y3, (x[ik] * w[..., None]).sum(1) / w.sum(1, keepdim=True), rtol=1e-4, atol=1e-5)


def test_decimation_indices_contract():
    with pytest.raises(ValueError):
        O.decimation_indices([0, 10], 0.5)
    idx, ptr = O.decimation_indices([0, 10, 13, 14], 4)
    assert ptr == [0, 2, 3, 4] and idx.numel() == 4  # max(1, n // 4) per cloud, never empty
    assert 0 <= idx[0] < 10 and 10 <= idx[2] < 13 and idx[3] == 13


def test_golden_vectors_pin_the_oracle():
    g = np.load(GOLDEN)
    net = O.RandLANetOracle(9, 6, return_logits=True)
    fill_params_deterministic(net, int(g["param_seed"]))
    dec = [torch.from_numpy(g[f"dec{i}"]) for i in range(4)]
    x, pos, ptr = torch.from_numpy(g["x"]), torch.from_numpy(g["pos"]), torch.from_numpy(g["ptr"])
    net.eval()
    rec = {}
    with torch.no_grad():
        out = net(x, pos, None, ptr, decimation_idx=dec, record=rec)
    assert torch.allclose(out, torch.from_numpy(g["logits_eval"]), rtol=1e-4, atol=1e-5)
    assert torch.equal(rec["block1.knn_idx"].to(torch.int32), torch.from_numpy(g["knn_idx_level1"]))
    net.train()
    out_t = net(x, pos, None, ptr, decimation_idx=dec, dropout_mask=torch.from_numpy(g["dropout_mask"]))
    loss = torch.nn.functional.cross_entropy(out_t, torch.from_numpy(g["y"]))
    loss.backward()
    assert torch.allclose(out_t, torch.from_numpy(g["logits_train"]), rtol=1e-3, atol=1e-4)
    assert abs(loss.item() - float(g["loss_train"])) < 1e-4
    # gradients: fp32 summation order changes with the thread count -> compare in relative L2
    for got, key in ((net.fc0.weight.grad, "grad_fc0_weight"),
                     (net.block4.lfa2.mlp_encoder.lins[0].weight.grad, "grad_block4_lfa2_enc"),
                     (net.block1.lfa1.mlp_attention.lins[0].weight.grad, "grad_block1_lfa1_att")):
        ref = torch.from_numpy(g[key])
        assert (got - ref).norm() / ref.norm() < 1e-2, key


def test_interpolator_reduce_by_hand():
    """interpolation.py:98-164 restated: overlapping predictions are summed, rows come back in stored order, entropy
    is the Shannon entropy of the softmax (natural log)."""
    from oracle.randla_oracle import interpolator_reduce

    a = torch.tensor([[1.0, 2.0, 0.5], [0.0, 0.0, 0.0]])
    b = torch.tensor([[0.5, -1.0, 3.0]])
    rows, probas, preds, entropy, idx = interpolator_reduce([a, b], [np.array([4, 2]), np.array([4])], nb_points=6)
    assert idx.tolist() == [4, 2, 4]
    merged = a[0] + b[0]
    assert torch.allclose(rows, torch.stack([merged, a[1], merged]))
    p = torch.exp(merged) / torch.exp(merged).sum()
    assert torch.allclose(probas[0], p, atol=1e-6) and torch.allclose(probas[2], p, atol=1e-6)
    assert torch.allclose(probas[1], torch.full((3,), 1 / 3), atol=1e-6)
    assert preds.tolist() == [2, 0, 2]                     # uniform row: first maximum
    assert abs(entropy[1].item() - np.log(3.0)) < 1e-6
    assert abs(entropy[0].item() + (p * p.log()).sum().item()) < 1e-6


def test_prep_oracle_grid_sampling_by_hand():
    """GridSampling restated (PyG voxel_grid + consecutive_cluster + scatter mean / label majority)."""
    from oracle import prep_oracle as P

    pos = torch.tensor([[0.0, 0.0, 0.0], [0.1, 0.1, 0.1], [0.3, 0.0, 0.0], [0.0, 0.3, 0.0], [0.2, 0.2, 0.2],
                        [0.26, 0.01, 0.01]])
    x = torch.arange(12, dtype=torch.float32).reshape(6, 2)
    y = torch.tensor([2, 1, 4, 0, 1, 3])
    p, xx, yy, inv = P.grid_sampling(pos, x, y, 0.25)
    # voxels (x fastest): (0,0,0) <- points 0,1,4 ; (1,0,0) <- points 2,5 ; (0,1,0) <- point 3
    assert inv.tolist() == [0, 0, 1, 2, 0, 1]
    assert torch.allclose(p[0], pos[[0, 1, 4]].mean(0)) and torch.allclose(p[1], pos[[2, 5]].mean(0))
    assert torch.allclose(xx[0], x[[0, 1, 4]].mean(0)) and torch.allclose(xx[2], x[3])
    assert yy.tolist() == [1, 3, 0]            # majority 1; tie between 4 and 3 -> first maximum = 3; single 0
    # node budget bookkeeping and the standardisation quirk (clamp bound = 3 * std of the raw channel)
    assert [P.budget_counts(n, 300, 40000) for n in (0, 1, 299, 300, 40000, 50000)] == [0, 300, 300, 300, 40000, 40000]
    c = P.minimum_num_nodes_choice(7, 20, torch.Generator().manual_seed(0))
    assert c.shape == (20,) and sorted(c[:7].tolist()) == list(range(7)) and sorted(c[7:14].tolist()) == list(range(7))
    ch = torch.tensor([0.0, 0.1, 0.2, 50.0])
    s = P.standardize_channel(ch)
    assert torch.allclose(s, (ch - ch.mean()) / (ch.std() + 1e-6))     # |z| < 3 * std here: no clamping
    assert torch.equal(P.standardize_channel(torch.tensor([5.0])), torch.tensor([0.0]))   # std NaN -> 1
    q = P.normalize_pos(P.nullify_lowest_z(P.center(pos)), 50)
    assert float(q[:, 2].min()) == 0.0 and torch.allclose(q[:, :2].mean(0), torch.zeros(2), atol=1e-7)


def test_golden_vectors_pin_the_prep_and_interpolator_oracles():
    """tests/golden/prep_small.npz (tests/golden/make_golden_prep.py): data preparation chain + merged predictions."""
    from oracle import prep_oracle as P
    from oracle.randla_oracle import interpolator_reduce

    g = np.load(os.path.join(os.path.dirname(GOLDEN), "prep_small.npz"))
    p, xx, yy, optr = P.prepare_tiles(torch.from_numpy(g["pos"]), torch.from_numpy(g["x"]), torch.from_numpy(g["y"]),
                                      g["ptr"].tolist(), 0.25, 50, 0, 7)
    assert optr == g["prep_ptr"].tolist()
    assert torch.allclose(p, torch.from_numpy(g["prep_pos"]), rtol=0, atol=1e-6)
    assert torch.allclose(xx, torch.from_numpy(g["prep_x"]), rtol=1e-5, atol=1e-5)
    assert torch.equal(yy, torch.from_numpy(g["prep_y"]))
    logits = [torch.from_numpy(g[f"logits{i}"]) for i in range(3)]
    idx = [g[f"idx{i}"] for i in range(3)]
    rows, probas, preds, entropy, cat_idx = interpolator_reduce(logits, idx, int(g["nb_points"]))
    assert torch.equal(cat_idx, torch.from_numpy(g["cat_idx"])) and torch.equal(rows, torch.from_numpy(g["rows"]))
    assert torch.allclose(probas, torch.from_numpy(g["probas"]), rtol=1e-6, atol=1e-7)
    assert torch.equal(preds, torch.from_numpy(g["preds"]))
    assert torch.allclose(entropy, torch.from_numpy(g["entropy"]), rtol=1e-6, atol=1e-6)
