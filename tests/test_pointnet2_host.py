"""CPU tests of the PointNet++ variant's host logic and of its oracle (no GPU)."""
import numpy as np
import torch


def test_fps_oracle_is_farthest_first():
    """The restated sampler against its definition, written differently (a full distance matrix in fp64)."""
    from oracle.pointnet2_oracle import fps_exact

    rs = np.random.RandomState(0)
    sizes, keep = [40, 1, 23], [10, 1, 23]
    pos = torch.from_numpy(rs.uniform(-1, 1, (sum(sizes), 3)).astype(np.float32))
    ptr = [0, 40, 41, 64]
    ptr_out = [0, 10, 11, 34]
    idx = fps_exact(pos, ptr, ptr_out, start=[3, 0, 22]).tolist()
    assert idx[0] == 3 and idx[10] == 40 and idx[11] == 41 + 22
    for b in range(3):
        sel = idx[ptr_out[b]:ptr_out[b + 1]]
        assert all(ptr[b] <= i < ptr[b + 1] for i in sel) and len(set(sel)) == len(sel)
        p = pos[ptr[b]:ptr[b + 1]].double()
        d = torch.cdist(p, p)
        chosen = [sel[0] - ptr[b]]
        for s in range(1, len(sel)):
            mind = d[:, chosen].min(dim=1).values
            # the next point realises the maximum of the min-distances (up to fp32 rounding of the oracle's arithmetic)
            assert mind[sel[s] - ptr[b]].item() >= mind.max().item() - 1e-6
            chosen.append(sel[s] - ptr[b])


def test_plan_and_parameter_tree():
    from myria3d_amd.pointnet2 import HipPointNet2, make_sa_plan
    from oracle.pointnet2_oracle import PointNet2Oracle, level_sizes

    plan = make_sa_plan([0, 50, 57, 187], 4, 16, "cpu")
    assert plan.sizes == [[50, 7, 130], [12, 1, 32], [3, 1, 8], [1, 1, 2]]
    assert [p.tolist() for p in plan.ptrs][1] == level_sizes([0, 50, 57, 187], 4)
    assert plan.num_edges == [12 * 16 + 7 + 32 * 16, 3 * 12 + 1 + 8 * 16, 3 + 1 + 2 * 8]
    assert plan.segs[0][:3].tolist() == [0, 16, 32] and plan.segs[0][12:15].tolist() == [192, 199, 215]
    net, ref = HipPointNet2(9, 6), PointNet2Oracle(9, 6)
    a = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert a == b
    assert a["sa1.nn.lins.0.weight"] == (32, 12) and a["fp1.nn.lins.0.weight"] == (64, 73)
    assert "sa3.nn.norms.2.module.running_var" in a and a["fc_classif.weight"] == (6, 32)


def test_oracle_runs_on_cpu_and_max_aggregation_routes_the_gradient():
    from oracle.pointnet2_oracle import PointNet2Oracle
    from tests._util import fill_params_deterministic, rand_batch

    ref = PointNet2Oracle(9, 6, num_neighbors=8, return_logits=False)
    fill_params_deterministic(ref, 1)
    x, pos, batch, ptr = rand_batch([60, 5, 33], seed=2)
    ref.train()
    out = ref(x, pos, batch, ptr)
    assert out.shape == (98, 6) and torch.allclose(out.exp().sum(1), torch.ones(98), atol=1e-5)
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in ref.parameters())
    assert [t.numel() for t in ref.last_sample_idx] == [15 + 1 + 8, 3 + 1 + 2, 1 + 1 + 1]


def test_oracle_reproduces_its_committed_fixture():
    """tests/golden/pointnet2_small.npz (make_golden_pointnet2.py): the restated oracle has not drifted.  The fixture pins
    the oracle to itself only — there is no reference implementation of this variant (model.py:12)."""
    import os

    from oracle.pointnet2_oracle import PointNet2Oracle
    from tests._util import fill_params_deterministic

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pointnet2_small.npz"))
    x, pos, ptr = torch.from_numpy(g["x"]), torch.from_numpy(g["pos"]), torch.from_numpy(g["ptr"])
    net = PointNet2Oracle(9, 6, num_neighbors=int(g["k"]), return_logits=True)
    fill_params_deterministic(net, int(g["param_seed"]))
    net.eval()
    with torch.no_grad():
        out = net(x, pos, None, ptr)
    for i in range(3):
        assert np.array_equal(net.last_sample_idx[i].numpy(), g[f"fps{i}"]), i
    assert np.allclose(out.numpy(), g["logits_eval"], rtol=1e-5, atol=1e-5)
    net.train()
    lt = net(x, pos, None, ptr, dropout_mask=torch.from_numpy(g["dropout_mask"]))
    loss = torch.nn.functional.cross_entropy(lt, torch.from_numpy(g["y"]))
    loss.backward()
    assert abs(loss.item() - float(g["loss_train"])) < 1e-5
    assert np.allclose(dict(net.named_parameters())["sa3.nn.lins.2.weight"].grad.numpy(), g["grad_sa3_lin2"], rtol=1e-3,
                       atol=1e-6)
