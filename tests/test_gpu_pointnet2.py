"""GPU parity of the PointNet++ set-abstraction variant (BASELINE.json configs[4]) against ``oracle/pointnet2_oracle.py``.

No reference implementation of this variant exists (model.py:12), so the oracle is a restatement of the published
operators and parity is UNPINNED; what these tests pin is HIP kernels == restatement:
  farthest-point sampling   bit-exact index lists
  grouping / max            bit-exact forward, exact transpose
  eval logits               |d| <= 2e-4 + 2e-4*|ref|
  train logits              |d| <= 2e-3 + 2e-3*|ref|;  parameter grads: relative L2 error <= 5e-3 vs an fp64 oracle run
"""
import numpy as np
import pytest
import torch

from tests._util import fill_params_deterministic, rand_batch

pytestmark = pytest.mark.gpu


def _report(name, got, ref, rtol, atol):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs()
    print(f"[parity] {name}: max_abs_err={err.max().item():.3e} ref_absmax={ref.abs().max().item():.3e}")
    bad = err > atol + rtol * ref.abs()
    assert not bool(bad.any()), f"{name}: {int(bad.sum())} / {bad.numel()} outside tol, max err {err.max().item():.3e}"


def _ptr(sizes):
    return torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64)


# ------------------------------------------------------------------------------------------
# farthest-point sampling
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sizes,keep,starts", [
    ([1], [1], None),
    ([2, 5], [2, 1], None),
    ([300, 211, 64], [75, 52, 16], None),
    ([300, 211, 64], [75, 52, 16], [17, 210, 0]),
    ([40, 40], [40, 40], None),            # keep everything: the order is still the farthest-first order
    ([5000], [1250], [4999]),              # 16 points per thread (positions in registers)
    ([12800, 9000], [3200, 2250], None),   # BASELINE tile size
    ([20000], [700], [123]),               # > 16 points per thread: positions re-read from L2
])
def test_fps_is_bit_exact(device, sizes, keep, starts):
    from myria3d_amd import ops
    from oracle.pointnet2_oracle import fps_exact

    rs = np.random.RandomState(sum(sizes))
    pos = torch.from_numpy(rs.uniform(-1, 1, (sum(sizes), 3)).astype(np.float32))
    ptr, ptr_out = _ptr(sizes), _ptr(keep)
    want = fps_exact(pos, ptr.tolist(), ptr_out.tolist(), starts)
    pos4 = ops.pad_pos(pos.to(device))
    st = torch.tensor(starts, dtype=torch.int32, device=device) if starts is not None else None
    got = ops.fps(pos4, ptr.to(device), ptr_out.to(device), int(ptr_out[-1]), max(sizes), st)
    assert torch.equal(got.cpu().long(), want)


@pytest.mark.parametrize("sizes,keep,starts", [
    ([20000, 2100, 17000], [5000, 525, 4250], None),   # a cloud below the bucket size in the batch
    ([40000], [10000], [777]),                         # the 40 000-point node budget: 625 buckets of 64 records
    ([16385, 30000], [4096, 7500], [5, 29999]),        # just above the register-resident sampler's range
    ([39999], [300], None),                            # a last bucket with one record missing
])
def test_fps_with_bucket_skipping_is_bit_exact(device, sizes, keep, starts):
    """fps_bucket_kernel (the points in the cell-sorted order of the kNN grid, buckets of 64 with bounding boxes, every bucket
    the new point cannot reach skipped): the index lists of the plain sampler and of the oracle, bit for bit."""
    from myria3d_amd import ops
    from oracle.pointnet2_oracle import fps_exact

    rs = np.random.RandomState(sum(sizes) + 1)
    pos = torch.from_numpy(rs.uniform(-1, 1, (sum(sizes), 3)).astype(np.float32))
    pos[:, 2] *= 0.3  # (flat clouds, like Lidar tiles)
    ptr, ptr_out = _ptr(sizes), _ptr(keep)
    pos4 = ops.pad_pos(pos.to(device))
    st = torch.tensor(starts, dtype=torch.int32, device=device) if starts is not None else None
    plain = ops.fps(pos4, ptr.to(device), ptr_out.to(device), int(ptr_out[-1]), max(sizes), st)
    ix = ops.KnnIndex(pos4, ptr.to(device))
    for _ in range(2):
        got = ops.fps(pos4, ptr.to(device), ptr_out.to(device), int(ptr_out[-1]), max(sizes), st, index=ix)
        assert torch.equal(got, plain), "bucket-skipping sampler vs the plain one"
    if sum(keep) <= 11000:  # (the pure-Python oracle loop is slow)
        want = fps_exact(pos, ptr.tolist(), ptr_out.tolist(), starts)
        assert torch.equal(got.cpu().long(), want)


def test_fps_on_duplicates_and_lattice_ties(device):
    """Integer lattice (many exactly equal distances) + repeated points: ties go to the smaller index on both sides."""
    from myria3d_amd import ops
    from oracle.pointnet2_oracle import fps_exact

    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(3), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    pos = torch.from_numpy(np.concatenate([g, g[:50], g[::-1][:30].copy()]))
    sizes = [pos.shape[0]]
    ptr, ptr_out = _ptr(sizes), _ptr([pos.shape[0]])  # more slots than distinct points: the tail re-selects at distance 0
    want = fps_exact(pos, ptr.tolist(), ptr_out.tolist())
    got = ops.fps(ops.pad_pos(pos.to(device)), ptr.to(device), ptr_out.to(device), sizes[0], sizes[0])
    assert torch.equal(got.cpu().long(), want)
    # more slots than distinct points on a cloud big enough for the bucket-skipping sampler (all minima reach 0; exactly
    # equal distances across bucket borders)
    g2 = np.stack(np.meshgrid(np.arange(32), np.arange(32), np.arange(12), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    pos3 = torch.from_numpy(np.concatenate([g2, g2[:6000]]))
    n3 = pos3.shape[0]
    p3 = ops.pad_pos(pos3.to(device))
    plain3 = ops.fps(p3, _ptr([n3]).to(device), _ptr([13000]).to(device), 13000, n3)
    got3 = ops.fps(p3, _ptr([n3]).to(device), _ptr([13000]).to(device), 13000, n3, index=ops.KnnIndex(p3, _ptr([n3]).to(device)))
    assert torch.equal(got3, plain3)
    want3 = fps_exact(pos3[:4000], [0, 4000], [0, 600])  # (the plain sampler itself against the oracle on a lattice)
    assert torch.equal(ops.fps(ops.pad_pos(pos3[:4000].to(device)), _ptr([4000]).to(device), _ptr([600]).to(device), 600, 4000).cpu().long(), want3)


def test_fps_spreads_points(device):
    """A property the domain offers at full size: the minimum pairwise distance of the FPS subset is at least that of a
    random subset of the same size (by a wide margin) — 16 x 12 800-point tiles, no oracle involved."""
    from myria3d_amd import ops
    from myria3d_amd.synthetic import synthetic_batch

    x, pos, batch, ptr, y = synthetic_batch([12800] * 4)
    keep = _ptr([3200] * 4)
    pos4 = ops.pad_pos(pos.to(device))
    idx = ops.fps(pos4, ptr.to(device), keep.to(device), int(keep[-1]), 12800).long()
    assert idx.numel() == 12800 and idx.unique().numel() == 12800
    for b in range(4):
        sel = idx[3200 * b:3200 * (b + 1)]
        assert int(sel.min()) >= 12800 * b and int(sel.max()) < 12800 * (b + 1) and int(sel[0]) == 12800 * b
        p = pos.to(device)[sel]
        d = torch.cdist(p, p) + torch.eye(3200, device=device) * 1e9
        rnd = pos.to(device)[12800 * b + torch.randperm(12800, device=device)[:3200]]
        dr = torch.cdist(rnd, rnd) + torch.eye(3200, device=device) * 1e9
        assert d.min().item() > 3 * dr.min().item()


# ------------------------------------------------------------------------------------------
# grouping and max aggregation
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [9, 64])
def test_group_and_max_match_torch(device, C):
    from myria3d_amd import ops
    from myria3d_amd.pointnet2 import make_sa_plan

    sizes, K = [50, 7, 130], 16  # the 7-point cloud has fewer than K points: 7 edges per centre there
    plan = make_sa_plan(_ptr(sizes).tolist(), 4, K, device, levels=1)
    rs = np.random.RandomState(C)
    n, m = sum(sizes), plan.totals[1]
    x = torch.from_numpy(rs.normal(size=(n, C)).astype(np.float32)).to(device).requires_grad_()
    pos = torch.from_numpy(rs.uniform(0, 1, (n, 3)).astype(np.float32)).to(device)
    pos4 = ops.pad_pos(pos)
    sel = ops.fps(pos4, plan.ptrs[0], plan.ptrs[1], m, max(sizes))
    ctr = ops.gather_rows(pos4, sel)
    nbr, _ = ops.KnnIndex(pos4, plan.ptrs[0]).query(K, pos_qry=ctr, ptr_qry=plan.ptrs[1])
    seg = plan.segs[0]
    E = plan.num_edges[0]
    assert E == 12 * 16 + 1 * 7 + 32 * 16
    ldo = (C + 6) // 4 * 4
    rows, esrc, ectr = ops.SAGroupFn.apply(x, pos4, ctr, nbr, seg, E, ldo)
    valid = nbr >= 0
    assert int(valid.sum()) == E
    j = nbr[valid].long()
    i = torch.arange(m, device=device)[:, None].expand(m, K)[valid]
    assert torch.equal(esrc.long(), j) and torch.equal(ectr.long(), i)
    xd = x.detach().clone().requires_grad_()
    want = torch.cat([xd[j], pos[j] - pos[sel.long()][i]], 1)
    assert torch.equal(rows[:, :C + 3], want) and bool((rows[:, C + 3:] == 0).all())
    # max over each centre's edges of a function of the rows, forward and backward
    wmat = torch.from_numpy(rs.normal(size=(C + 3, 24)).astype(np.float32)).to(device)
    out = ops.SegMaxFn.apply(rows[:, :C + 3] @ wmat, seg, ectr, m)
    dense = torch.full((m, K, 24), float("-inf"), device=device)
    dense[valid] = want @ wmat
    ref = dense.max(dim=1).values
    assert torch.equal(out, ref)
    gout = torch.from_numpy(rs.normal(size=(m, 24)).astype(np.float32)).to(device)
    out.backward(gout)
    ref.backward(gout)
    _report("dx", x.grad, xd.grad, 1e-5, 1e-5)


# ------------------------------------------------------------------------------------------
# the net
# ------------------------------------------------------------------------------------------
def _pair(device, num_features=9, num_classes=6, k=16, seed=0, **kw):
    from myria3d_amd.pointnet2 import HipPointNet2
    from oracle.pointnet2_oracle import PointNet2Oracle

    ref = PointNet2Oracle(num_features, num_classes, num_neighbors=k, return_logits=True)
    fill_params_deterministic(ref, seed)
    net = HipPointNet2(num_features, num_classes, num_neighbors=k, return_logits=True, **kw)
    net.load_state_dict(ref.state_dict())  # strict: the two parameter trees have the same keys and shapes
    return ref, net.to(device)


@pytest.mark.parametrize("sizes,k", [([300, 211], 16), ([64, 700, 20], 32), ([1250, 1000], 16), ([5, 1, 40], 8)])
def test_eval_logits_match_oracle(device, sizes, k):
    ref, net = _pair(device, k=k, seed=len(sizes))
    x, pos, batch, ptr = rand_batch(sizes, seed=sum(sizes))
    ref.eval(), net.eval()
    rec_r, rec_g = {}, {}
    with torch.no_grad():
        out_r = ref(x, pos, batch, ptr, record=rec_r)
        out_g = net(x.to(device), pos.to(device), batch.to(device), ptr.to(device), record=rec_g)
    for lvl in range(3):  # the HIP sampler chose the oracle's points
        assert torch.equal(net.last_sample_idx[lvl].cpu().long(), ref.last_sample_idx[lvl]), lvl
    for key in sorted(rec_g):
        _report(key, rec_g[key], rec_r[key], 2e-4, 2e-4)
    _report("logits", out_g, out_r, 2e-4, 2e-4)
    assert (out_g.cpu().argmax(1) == out_r.argmax(1)).float().mean().item() >= 0.999


def test_forward_contract(device):
    """The reference net's surface (pyg_randla_net.py:23-30,55-88): x=None uses pos, log-softmax unless return_logits,
    ValueError for a decimation factor below 1, row order preserved, random subsampling as an option."""
    from myria3d_amd.pointnet2 import HipPointNet2

    x, pos, batch, ptr = rand_batch([120, 77], num_features=3, seed=5)
    net = HipPointNet2(3, 7, num_neighbors=8, subsampling="random").to(device).eval()
    with torch.no_grad():
        out = net(None, pos.to(device), batch.to(device), ptr.to(device))
        assert out.shape == (197, 7) and torch.allclose(out.exp().sum(1), torch.ones(197, device=device), atol=1e-4)
        a = net.last_sample_idx[0].clone()
        net(None, pos.to(device), batch.to(device), ptr.to(device))
        assert not torch.equal(a, net.last_sample_idx[0])  # a fresh random subset per forward
        assert a.unique().numel() == a.numel() and int(a[:30].max()) < 120 and int(a[30:].min()) >= 120
    with pytest.raises(ValueError):
        HipPointNet2(3, 7, decimation=0).to(device)(None, pos.to(device), batch.to(device), ptr.to(device))
    with pytest.raises(ValueError):
        HipPointNet2(3, 7, subsampling="voxel")


def test_prefetched_geometry_gives_the_same_step(device):
    """Round 5: ``HipPointNet2.prefetch_geometry`` computes the position-only tables of the NEXT batch (farthest-point sampling,
    grids, grouping and 1-NN tables) on a side stream; the forward that picks them up must give exactly what the forward
    that computes them inline gives — logits, loss gradients, the sampled index lists — and a prefetch for ANOTHER batch, or
    a ``pos`` written since, must be dropped."""
    from myria3d_amd.pointnet2 import HipPointNet2

    xa, pa, ba, ptra = rand_batch([900, 640, 77], seed=3)
    xb, pb, bb, ptrb = rand_batch([500, 1200], seed=4)
    A = tuple(t.to(device) for t in (xa, pa, ba, ptra))
    B = tuple(t.to(device) for t in (xb, pb, bb, ptrb))
    torch.manual_seed(0)
    net = HipPointNet2(9, 6, num_neighbors=16, return_logits=True).to(device).train()
    ya = torch.from_numpy(np.random.RandomState(2).randint(0, 6, (A[0].shape[0],))).to(device)

    def run(prefetch):
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.momentum = 0.0
        net.zero_grad(set_to_none=True)
        if prefetch:
            net.prefetch_geometry(A[1], A[3])
            assert net._look is not None
        out = net(*A, dropout_mask=torch.ones(A[0].shape[0], 32, device=device))
        assert net._look is None
        torch.nn.functional.cross_entropy(out, ya).backward()
        torch.cuda.synchronize()
        return out.detach().clone(), [s.clone() for s in net.last_sample_idx], [p.grad.clone() for p in net.parameters()]

    o0, s0, g0 = run(False)
    o1, s1, g1 = run(True)
    assert all(torch.equal(a, b) for a, b in zip(s0, s1)), "same sampled points"
    assert torch.equal(o0, o1), "same logits, bit for bit (the same kernels on the same tables)"
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * max(1.0, a.abs().max().item()))
    # a prefetch for another batch is not used by this one (and stays queued for its own forward); one whose positions were
    # written afterwards is dropped
    net.prefetch_geometry(B[1], B[3])
    with torch.no_grad():
        o2 = net(*A, dropout_mask=torch.ones(A[0].shape[0], 32, device=device))
    assert torch.equal(o2, o0) and net._look is not None and net._look[0][0] is B[1]
    with torch.no_grad():
        net(*B)
    assert net._look is None
    # two batches queued, consumed first in, first out
    net.prefetch_geometry(A[1], A[3])
    net.prefetch_geometry(B[1], B[3], wait_main=False)
    with torch.no_grad():
        o3 = net(*A, dropout_mask=torch.ones(A[0].shape[0], 32, device=device))
        assert torch.equal(o3, o0) and len(net._look) == 1
        net(*B)
    assert net._look is None
    # round 6: three batches in flight, each on its own stream pair (prefetch_depth = 3): the same batch queued three times is
    # consumed three times, oldest first, with identical results; a fourth entry pushes the oldest out
    assert net.prefetch_depth == 3
    for _ in range(3):
        net.prefetch_geometry(A[1], A[3], wait_main=False)
    assert len(net._look) == 3 and len(net._sides) == 3
    with torch.no_grad():
        for left in (2, 1, 0):
            o4 = net(*A, dropout_mask=torch.ones(A[0].shape[0], 32, device=device))
            assert torch.equal(o4, o0) and len(net._look or []) == left
    pa2 = A[1].clone()
    net.prefetch_geometry(pa2, A[3])
    pa2.mul_(1.0)  # (bumps the version counter)
    with torch.no_grad():
        net(A[0], pa2, A[2], A[3], dropout_mask=torch.ones(A[0].shape[0], 32, device=device))
    assert net._look is None


@pytest.mark.parametrize("sizes,k", [([300, 211], 16), ([64, 700, 20], 32)])
def test_train_forward_backward_match_fp64_oracle(device, sizes, k):
    x, pos, batch, ptr = rand_batch(sizes, seed=sizes[0])
    y = torch.from_numpy(np.random.RandomState(1).randint(0, 6, (sum(sizes),)))
    _train_parity(device, x, pos, batch, ptr, y, k, inject_samples=False)


def test_dense_tile_40000_points_k32_train_matches_fp64_oracle(device):
    """BASELINE configs[4] at full tile size, TRAIN mode: forward, cross-entropy, backward of the HIP net against the fp64
    oracle on one 40 000-point tile with K = 32 (the oracle takes the sampled indices of the HIP net — the sampler itself is
    compared bit for bit at this size in the eval test — so the test does not spend 10 000 numpy FPS iterations twice)."""
    from myria3d_amd.synthetic import synthetic_batch

    x, pos, batch, ptr, y = synthetic_batch([40000])
    _train_parity(device, x, pos, batch, ptr, y, 32, inject_samples=True)


def _train_parity(device, x, pos, batch, ptr, y, k, inject_samples):
    ref, net = _pair(device, k=k, seed=11)
    ref = ref.double()
    n = x.shape[0]
    mask = torch.from_numpy((np.random.RandomState(2).uniform(size=(n, 32)) > 0.5).astype(np.float32))
    ref.train(), net.train()
    out_g = net(x.to(device), pos.to(device), batch.to(device), ptr.to(device), dropout_mask=mask.to(device))
    loss_g = torch.nn.functional.cross_entropy(out_g, y.to(device))
    loss_g.backward()
    extra = dict(sample_idx=[s_.cpu() for s_ in net.last_sample_idx]) if inject_samples else {}
    out_r = ref(x.double(), pos.double(), batch, ptr, dropout_mask=mask.double(), **extra)
    loss_r = torch.nn.functional.cross_entropy(out_r, y)
    loss_r.backward()
    _report("train.logits", out_g, out_r, 2e-3, 2e-3)
    assert abs(loss_g.item() - loss_r.item()) < 1e-3 * max(1.0, abs(loss_r.item()))
    ref_params = dict(ref.named_parameters())
    worst = ("", 0.0)
    for name, p in net.named_parameters():
        assert p.grad is not None, f"{name} got no gradient"
        gr, gg = ref_params[name].grad.double(), p.grad.detach().cpu().double()
        if ".lins." in name and name.endswith("bias"):  # Linear bias in front of a train-mode BatchNorm: analytically zero
            assert gg.abs().max().item() < 1e-5 and gr.abs().max().item() < 1e-6, name
            continue
        if gr.norm().item() < 1e-8:
            # analytically zero as well: when every group maximum of a level sits on the positive side of the LeakyReLU,
            # the last BatchNorm's shift passes linearly through the max into train-mode BatchNorms, which remove it
            assert gg.abs().max().item() < 1e-5, name
            continue
        rel = (gg - gr).norm().item() / max(gr.norm().item(), 1e-12)
        worst = max(worst, (name, rel), key=lambda t: t[1])
        assert rel <= 5e-3, f"grad {name}: relative L2 error {rel:.3e}"
    print(f"[parity] worst parameter-gradient relative L2 error: {worst[0]} {worst[1]:.3e}")
    got_buffers = dict(net.named_buffers())
    for nr, br in ref.named_buffers():
        if nr.endswith("running_mean") or nr.endswith("running_var"):
            assert torch.allclose(got_buffers[nr].cpu().double(), br, rtol=1e-3, atol=1e-5), nr


def test_eval_forward_is_differentiable(device):
    ref, net = _pair(device, k=8, seed=3)
    x, pos, batch, ptr = rand_batch([90, 61], seed=9)
    ref.eval(), net.eval()
    xr = x.clone().requires_grad_()
    xg = x.to(device).requires_grad_()
    ref(xr, pos, batch, ptr).square().sum().backward()
    net(xg, pos.to(device), batch.to(device), ptr.to(device)).square().sum().backward()
    _report("dx", xg.grad, xr.grad, 2e-3, 2e-4)
    gr = dict(ref.named_parameters())["sa2.nn.lins.1.weight"].grad
    _report("dW", dict(net.named_parameters())["sa2.nn.lins.1.weight"].grad, gr, 2e-3, 2e-3 * gr.abs().max().item())


def test_dense_tile_40000_points_k32_eval_matches_oracle(device):
    """BASELINE configs[4] at full tile size (one tile): FPS of 10 000 / 2 500 / 625 points, K = 32 grouping."""
    from myria3d_amd.synthetic import synthetic_batch

    ref, net = _pair(device, k=32, seed=2)
    x, pos, batch, ptr, y = synthetic_batch([40000])
    ref.eval(), net.eval()
    with torch.no_grad():
        out_g = net(x.to(device), pos.to(device), batch.to(device), ptr.to(device))
        out_r = ref(x, pos, batch, ptr, sample_idx=[s.cpu() for s in net.last_sample_idx])
    # (the sampler itself is compared at this size without the net: 10 000 numpy iterations over 40 000 points)
    from oracle.pointnet2_oracle import fps_exact
    want = fps_exact(pos, [0, 40000], [0, 10000])
    assert torch.equal(net.last_sample_idx[0].cpu().long(), want)
    _report("logits", out_g, out_r, 2e-4, 2e-4)


def test_committed_fixture_vs_hip_net(device):
    """The HIP net against tests/golden/pointnet2_small.npz directly (no oracle run on the GPU box): sampled indices
    bit-exact, eval / train logits, loss, four parameter gradients, a running variance."""
    import os

    from myria3d_amd.pointnet2 import HipPointNet2
    from oracle.pointnet2_oracle import PointNet2Oracle

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pointnet2_small.npz"))
    x, pos, ptr = (torch.from_numpy(g[k]).to(device) for k in ("x", "pos", "ptr"))
    shell = PointNet2Oracle(9, 6, num_neighbors=int(g["k"]), return_logits=True)  # (parameter tree + deterministic fill only)
    fill_params_deterministic(shell, int(g["param_seed"]))
    net = HipPointNet2(9, 6, num_neighbors=int(g["k"]), return_logits=True)
    net.load_state_dict(shell.state_dict())
    net = net.to(device).eval()
    with torch.no_grad():
        out = net(x, pos, None, ptr)
    for i in range(3):
        assert np.array_equal(net.last_sample_idx[i].cpu().numpy().astype(np.int64), g[f"fps{i}"]), i
    _report("fixture.logits_eval", out, torch.from_numpy(g["logits_eval"]), 2e-4, 2e-4)
    net.train()
    lt = net(x, pos, None, ptr, dropout_mask=torch.from_numpy(g["dropout_mask"]).to(device))
    loss = torch.nn.functional.cross_entropy(lt, torch.from_numpy(g["y"]).to(device))
    loss.backward()
    _report("fixture.logits_train", lt, torch.from_numpy(g["logits_train"]), 2e-3, 2e-3)
    assert abs(loss.item() - float(g["loss_train"])) < 1e-3
    params = dict(net.named_parameters())
    for name, key in (("sa1.nn.lins.0.weight", "grad_sa1_lin0"), ("sa3.nn.lins.2.weight", "grad_sa3_lin2"),
                      ("fp1.nn.lins.0.weight", "grad_fp1_lin0"), ("fc_classif.weight", "grad_fc_classif")):
        gr = torch.from_numpy(g[key]).double()
        rel = (params[name].grad.cpu().double() - gr).norm().item() / gr.norm().item()
        assert rel <= 5e-3, (name, rel)  # (fp32 on both sides)
    assert torch.allclose(net.sa2.nn.norms[1].module.running_var.cpu(), torch.from_numpy(g["running_var_sa2_bn1"]),
                          rtol=1e-3, atol=1e-5)
