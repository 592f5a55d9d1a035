"""GPU parity tests, op by op: every HIP kernel reached through the C ABI vs the CPU oracle / plain torch fp32-fp64.

Tolerances are written next to each check.  Integer/index work (kNN, gathers, decimation) is bit-exact.
"""
import numpy as np
import pytest
import torch

from tests._util import fill_params_deterministic, rand_batch

pytestmark = pytest.mark.gpu


def _close(name, got, ref, rtol, atol):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    worst = (err - tol).max().item() if err.numel() else 0.0
    print(f"[parity] {name}: max_abs_err={err.max().item() if err.numel() else 0:.3e} ref_max={ref.abs().max().item() if ref.numel() else 0:.3e}")
    assert worst <= 0, f"{name}: max abs err {err.max().item():.3e} exceeds tol (rtol={rtol}, atol={atol})"


# ----------------------------------------------------------------------------------------------- kNN
def _knn_case(device, pos, ptr, k):
    from myria3d_amd import ops
    from oracle.randla_oracle import knn_exact

    ref_idx, ref_d2 = knn_exact(pos, ptr.tolist(), pos, ptr.tolist(), k)
    index = ops.KnnIndex(pos.to(device), ptr.to(device))
    idx, d2 = index.query(k, qry=index, want_d2=True)
    assert torch.equal(idx.cpu().long(), ref_idx), "self-kNN indices differ from the oracle"
    assert torch.equal(d2.cpu(), ref_d2), "self-kNN squared distances are not bit-identical"
    # row-order queries (qmode 0) must give the same table
    idx2, _ = index.query(k, pos_qry=pos.to(device), ptr_qry=ptr.to(device))
    assert torch.equal(idx2.cpu().long(), ref_idx)


@pytest.mark.parametrize("sizes", [[300, 211], [50, 50], [3000], [1, 2, 17, 5], [1]])
@pytest.mark.parametrize("k", [16, 1, 10])
def test_knn_self_bit_exact(device, sizes, k):
    _, pos, _, ptr = rand_batch(sizes, seed=len(sizes) + k)
    _knn_case(device, pos, ptr, k)


def test_knn_lidar_tile_and_duplicates(device):
    from oracle.randla_oracle import synthetic_batch

    _, pos, _, ptr, _ = synthetic_batch([2500, 1800])
    _knn_case(device, pos, ptr, 16)
    # duplicated points (MinimumNumNodes duplicates points: myria3d/pctl/transforms/transforms.py:74-77)
    pos_dup = torch.cat([pos[:40].repeat(8, 1), pos[:200]])
    _knn_case(device, pos_dup, torch.tensor([0, pos_dup.shape[0]]), 16)
    # degenerate: all points identical / collinear in z
    same = torch.zeros(100, 3) + 0.25
    _knn_case(device, same, torch.tensor([0, 100]), 16)
    line = torch.zeros(300, 3)
    line[:, 2] = torch.linspace(0, 1, 300)
    _knn_case(device, line, torch.tensor([0, 300]), 16)


def test_knn_deferred_insertion_kernel_is_bit_identical(device):
    """The deferred-insertion kernel (circular rings, sorting-network drains; what the level-1 and K = 32 launches run) forced
    on small inputs (``kernel="queue"``) must give the oracle's table bit for bit: ragged clouds, clouds smaller than K,
    duplicates, collinear points, K = 8 / 10 / 16 / 32 / 50, a dense cluster inside a sparse cloud, cell-sorted output,
    queries of another point set."""
    from myria3d_amd import ops
    from oracle.randla_oracle import knn_exact, synthetic_batch

    def case(pos, ptr, k):
        ref_idx, ref_d2 = knn_exact(pos, ptr.tolist(), pos, ptr.tolist(), k)
        ix = ops.KnnIndex(pos.to(device), ptr.to(device))
        for kern in ("queue", "direct"):
            idx, d2 = ix.query(k, qry=ix, want_d2=True, kernel=kern)
            assert torch.equal(idx.cpu().long(), ref_idx), (kern, k)
            assert torch.equal(d2.cpu(), ref_d2), (kern, k)
        idx2, _ = ix.query(k, pos_qry=pos.to(device), ptr_qry=ptr.to(device), kernel="queue")  # row-order queries
        assert torch.equal(idx2.cpu().long(), ref_idx)

    for sizes, k in (([300, 211], 16), ([1, 2, 17, 5], 16), ([3000], 8), ([50, 50], 32), ([700, 450], 32), ([9000, 64], 16),
                     ([2000], 10), ([800, 30], 50)):
        _, pos, _, ptr = rand_batch(sizes, seed=len(sizes) + k)
        case(pos, ptr, k)
    _, pos, _, ptr, _ = synthetic_batch([2500, 1800, 4000])
    case(pos, ptr, 16)
    pos_dup = torch.cat([pos[:40].repeat(8, 1), pos[:200]])
    case(pos_dup, torch.tensor([0, pos_dup.shape[0]]), 16)
    line = torch.zeros(300, 3)
    line[:, 2] = torch.linspace(0, 1, 300)
    case(line, torch.tensor([0, 300]), 16)
    same = torch.zeros(100, 3) + 0.25
    case(same, torch.tensor([0, 100]), 16)
    # a cluster far denser than its surroundings + a sparse rest (the disc of the cluster's queries is tiny, the others' wide)
    dense = torch.cat([torch.rand(1500, 3) * 0.01 + 0.5, torch.rand(2000, 3)])
    case(dense, torch.tensor([0, 3500]), 16)
    # un-normalised coordinates (metres, Lambert-93-sized offsets): the trimming slack must scale with the magnitudes
    big = torch.rand(3000, 3) * torch.tensor([50.0, 50.0, 20.0]) + torch.tensor([843000.0, 6519000.0, 200.0])
    case(big, torch.tensor([0, 3000]), 16)
    # cell-sorted io + a different source set
    sub = torch.cat([torch.randperm(2500)[:600], 2500 + torch.randperm(1800)[:450], 4300 + torch.randperm(4000)[:1000]])
    ptr_s = torch.tensor([0, 600, 1050, 2050])
    src = pos[sub].contiguous()
    ref_idx, ref_d2 = knn_exact(src, ptr_s.tolist(), pos, ptr.tolist(), 8)
    si, qi = ops.KnnIndex(src.to(device), ptr_s.to(device)), ops.KnnIndex(pos.to(device), ptr.to(device))
    idx, d2 = si.query(8, qry=qi, want_d2=True, kernel="queue")
    assert torch.equal(idx.cpu().long(), ref_idx) and torch.equal(d2.cpu(), ref_d2)
    idx_s, _ = si.query(8, qry=qi, sorted_io=True, kernel="queue")  # rows = slots of qi, ids = slots of si
    back = si.perm.long()[idx_s.long()][qi.inv.long()]
    assert torch.equal(back.cpu(), ref_idx)


def test_knn_two_kernels_agree_at_full_size(device):
    """BASELINE config 2 and config 5 shapes (16 x 12 800, K = 16; 4 x 40 000, K = 32): the deferred-insertion kernel (the
    default at these sizes) against the direct-insertion kernel — equal tables and distances; one tile of each against the
    oracle."""
    from myria3d_amd import ops
    from oracle.randla_oracle import knn_exact, synthetic_batch

    for sizes, k in (([12800] * 16, 16), ([40000] * 4, 32)):
        _, pos, _, ptr, _ = synthetic_batch(sizes)
        ix = ops.KnnIndex(pos.to(device), ptr.to(device))
        ref, ref_d2 = ix.query(k, qry=ix, want_d2=True, sorted_io=True, kernel="direct")
        got, got_d2 = ix.query(k, qry=ix, want_d2=True, sorted_io=True)
        assert torch.equal(got, ref) and torch.equal(got_d2, ref_d2)
        n0 = sizes[0]
        o_idx, o_d2 = knn_exact(pos[:n0], [0, n0], pos[:n0], [0, n0], k)
        plain, plain_d2 = ix.query(k, qry=ix, want_d2=True)
        assert torch.equal(plain[:n0].cpu().long(), o_idx) and torch.equal(plain_d2[:n0].cpu(), o_d2)


@pytest.mark.parametrize("k", [65, 100])
def test_knn_up_to_the_upstream_limit_of_100(device, k):
    """torch_cluster's CUDA kNN asserts k <= 100: 64 < k <= 100 runs as two passes (the 64 nearest, then the
    next k - 64 in the same total order) and gives the oracle's table bit for bit, including clouds with fewer than k and
    fewer than 64 points; k = 101 is refused."""
    from myria3d_amd import ops
    from myria3d_amd._lib import M3DError
    from oracle.randla_oracle import knn_exact

    _, pos, _, ptr = rand_batch([400, 90, 30, 1500], seed=k)
    ref_idx, ref_d2 = knn_exact(pos, ptr.tolist(), pos, ptr.tolist(), k)
    ix = ops.KnnIndex(pos.to(device), ptr.to(device))
    idx, d2 = ix.query(k, qry=ix, want_d2=True)
    assert torch.equal(idx.cpu().long(), ref_idx) and torch.equal(d2.cpu(), ref_d2)
    idx_r, _ = ix.query(k, pos_qry=pos.to(device), ptr_qry=ptr.to(device))
    assert torch.equal(idx_r.cpu().long(), ref_idx)
    idx_s, _ = ix.query(k, qry=ix, sorted_io=True)
    back = torch.where(idx_s >= 0, ix.perm.long()[idx_s.long().clamp(min=0)], torch.full_like(idx_s, -1).long())[ix.inv.long()]
    assert torch.equal(back.cpu(), ref_idx)
    with pytest.raises(M3DError):
        ix.query(101, qry=ix)


def test_batched_queries_match_the_per_level_launches(device):
    """m3d_knn_query_batch / m3d_lfa_moments_batch: the four levels in one launch give the tables of four launches bit for
    bit (moments: same fp64 sums up to the order of the atomics)."""
    from myria3d_amd import ops
    from oracle.randla_oracle import synthetic_batch

    _, pos, _, ptr, _ = synthetic_batch([2500, 1800, 3100])
    pos, ptr = pos.to(device), ptr.to(device)
    idxs = [ops.KnnIndex(ops.pad_pos(pos), ptr)]
    ptrs = [ptr]
    for l in range(3):  # three decimated levels (every 4th point of each cloud)
        sizes = (ptrs[-1][1:] - ptrs[-1][:-1]) // 4
        keep = torch.cat([torch.arange(int(n), device=device) * 4 + int(ptrs[-1][b]) for b, n in enumerate(sizes)]).to(torch.int32)
        p_next = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(sizes, 0)])
        idxs.append(ops.KnnIndex(ops.gather_rows(idxs[-1].sorted_pos4, ops.gather_i32(idxs[-1].inv, keep)), p_next))
        ptrs.append(p_next)
    for k, pairs in ((16, [(ix, ix) for ix in idxs]), (1, [(idxs[l + 1], idxs[l]) for l in range(3)]),
                     (8, [(idxs[0], idxs[0])])):
        got = ops.knn_query_batch(pairs, k)
        for (src, qry), g in zip(pairs, got):
            ref, _ = src.query(k, qry=qry, sorted_io=True)
            assert torch.equal(g, ref), f"batched k={k} table differs from the single launch"
    tabs = ops.knn_query_batch([(ix, ix) for ix in idxs], 16)
    moms = ops.lfa_moments_batch([ix.sorted_pos4 for ix in idxs], tabs)
    for ix, t, m in zip(idxs, tabs, moms):
        ref = ops.lfa_moments(ix.sorted_pos4, t)
        assert m.shape == (65,) and m.data_ptr() % 16 == 0
        _close("moments(batch)", m, ref, 1e-12, 1e-12)


def test_knn_k32_and_cross_set(device):
    from myria3d_amd import ops
    from oracle.randla_oracle import knn_exact

    _, pos, _, ptr = rand_batch([700, 450], seed=3)
    _knn_case(device, pos, ptr, 32)
    # 1-NN / 3-NN of every point among a random subset (decoder upsampling pattern)
    sub = torch.cat([torch.randperm(700)[:175], 700 + torch.randperm(450)[:112]])
    ptr_s = torch.tensor([0, 175, 287])
    src = pos[sub].contiguous()
    for k in (1, 3):
        ref_idx, ref_d2 = knn_exact(src, ptr_s.tolist(), pos, ptr.tolist(), k)
        si = ops.KnnIndex(src.to(device), ptr_s.to(device))
        qi = ops.KnnIndex(pos.to(device), ptr.to(device))
        idx, d2 = si.query(k, qry=qi, want_d2=True)
        assert torch.equal(idx.cpu().long(), ref_idx)
        assert torch.equal(d2.cpu(), ref_d2)


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,K,N", [(1000, 9, 32), (777, 32, 4), (513, 64, 128), (300, 768, 256), (100, 32, 6),
                                   (65, 10, 8), (2, 512, 512), (20000, 32, 32), (5000, 64, 64), (4099, 16, 16),
                                   (3000, 160, 32), (1000, 256, 512), (70000, 8, 8), (33, 48, 20)])
def test_gemm_forward(device, M, K, N):
    from myria3d_amd import ops

    rs = np.random.RandomState(M + K + N)
    a = torch.from_numpy(rs.uniform(-1, 1, (M, K)).astype(np.float32))
    w = torch.from_numpy(rs.uniform(-1, 1, (N, K)).astype(np.float32))
    b = torch.from_numpy(rs.uniform(-1, 1, (N,)).astype(np.float32))
    sc = torch.from_numpy(rs.uniform(0.5, 1.5, (N,)).astype(np.float32))
    sh = torch.from_numpy(rs.uniform(-1, 1, (N,)).astype(np.float32))
    ref = a.double() @ w.double().t() + b.double()
    stats = ops.stat_buffer(M, N, K, device)
    stats.fill_(float("nan"))  # must be fully overwritten
    got = ops.gemm(a.to(device), w.to(device), M, N, K, bias=b.to(device), stats=stats)
    stats = stats.sum(0)
    # fp32 MFMA == fmaf chain; error bound ~ K * eps * sum|a||w|
    _close("gemm", got, ref, 1e-5, 2e-6 * K)
    # the statistics are those of the fp32 outputs: allow M * eps_f32 * |z|max (and its square) of rounding
    zmax = ref.abs().max().item()
    _close("gemm.stat_sum", stats[0], ref.sum(0), 1e-6, 2e-7 * M * zmax + 1e-5)
    _close("gemm.stat_sumsq", stats[1], (ref * ref).sum(0), 1e-6, 4e-7 * M * zmax * zmax + 1e-5)
    got2 = ops.gemm(a.to(device), w.to(device), M, N, K, bias=b.to(device), scale=sc.to(device), shift=sh.to(device),
                    act=True)
    ref2 = torch.nn.functional.leaky_relu(ref * sc.double() + sh.double(), 0.2)
    _close("gemm.affine_lrelu", got2, ref2, 1e-5, 4e-6 * K)



def test_csr_inverse_and_gather_sum_equal_the_atomic_scatter(device):
    """``m3d_csr_invert_batch`` + ``m3d_gather_sum_rows`` (backward of the decoder's ``x[nn]`` gathers, pyg_randla_net.py:250):
    the lists partition the mapped rows, and summing rows per target gives what the atomic ``m3d_scatter_add_rows`` gives —
    targets nobody maps to get zeros (no zero fill needed), unmapped rows (negative ids) are left out, an existing
    buffer can be added to."""
    from myria3d_amd import ops

    rs = np.random.RandomState(3)
    sizes = [(204800, 51200, 32), (51200, 12800, 128), (3200, 800, 512), (1000, 7, 8), (5, 4000, 4)]
    idxs = []
    for n, m, _ in sizes:
        ix = rs.randint(0, m, n).astype(np.int32)
        ix[rs.rand(n) < 0.01] = -1
        idxs.append(torch.from_numpy(ix).to(device))
    pairs = ops.csr_invert_batch(idxs, [m for _, m, _ in sizes])
    for (n, m, C), ix, (ptr, inv) in zip(sizes, idxs, pairs):
        ptr_h, inv_h, ix_h = ptr.cpu().numpy(), inv.cpu().numpy(), ix.cpu().numpy()
        cnt = np.bincount(ix_h[ix_h >= 0], minlength=m)
        assert ptr_h[0] == 0 and np.array_equal(np.diff(ptr_h), cnt)
        used = inv_h[:ptr_h[-1]]
        assert np.array_equal(np.sort(used), np.nonzero(ix_h >= 0)[0])          # every mapped row exactly once
        assert np.array_equal(ix_h[used], np.repeat(np.arange(m), cnt))          # ... in its target's list
        src = torch.from_numpy(rs.uniform(-1, 1, (n, C)).astype(np.float32)).to(device)
        ref = ops.scatter_add_rows(src, ix, m, out=torch.zeros(m, C, device=device))
        got = ops.gather_sum_rows(src, ptr, inv, m)
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5), (got - ref).abs().max().item()
        base = torch.from_numpy(rs.uniform(-1, 1, (m, C)).astype(np.float32)).to(device)
        got2 = ops.gather_sum_rows(src, ptr, inv, m, out=base.clone())
        assert torch.allclose(got2, ref + base, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,K,C", [(204800, 16, 8), (204800, 16, 4), (5000, 16, 8), (4097, 16, 4), (300, 16, 8), (70000, 32, 16), (1, 1, 4)])
def test_knn_reverse_lists_and_the_four_lane_gather(device, n, K, C):
    """Round 5: ``m3d_knn_reverse`` (one pass of atomics that keeps every edge's rank, multi-workgroup scan, atomic-free fill)
    + ``m3d_gather_sum_rows(long lists)``: every edge exactly once in the list of the point it names, padding (-1) left out,
    points nobody names get empty lists / zero rows, and the sums equal the atomic scatter's."""
    from myria3d_amd import ops

    rs = np.random.RandomState(n % 1000 + K)
    # a K-NN-like table: neighbours a few hundred rows around the centre, some rows never named, some padding
    ix = (np.arange(n)[:, None] + rs.randint(-300, 300, (n, K))) % max(n - n // 50, 1)
    ix[rs.rand(n, K) < 0.002] = -1
    ix = ix.astype(np.int32)
    idx = torch.from_numpy(ix).to(device)
    ptr, inv, slot = ops.knn_reverse(idx)
    ptr_h, inv_h, flat = ptr.cpu().numpy(), inv.cpu().numpy(), ix.reshape(-1)
    cnt = np.bincount(flat[flat >= 0], minlength=n)
    assert ptr_h[0] == 0 and np.array_equal(np.diff(ptr_h), cnt)
    used = inv_h[:ptr_h[-1]]
    assert np.array_equal(np.sort(used), np.nonzero(flat >= 0)[0])
    assert np.array_equal(flat[used], np.repeat(np.arange(n), cnt))
    slot_h = slot.cpu().numpy()
    assert np.array_equal(slot_h < 0, flat < 0)                                   # padding is in no list
    assert np.array_equal(inv_h[slot_h[flat >= 0]], np.nonzero(flat >= 0)[0])    # slot[e] is where inv holds e
    src = torch.from_numpy(rs.uniform(-1, 1, (n * K, C)).astype(np.float32)).to(device)
    ref = ops.scatter_add_rows(src, idx.view(-1), n, out=torch.zeros(n, C, device=device))
    got = ops.gather_sum_rows(src, ptr, inv, n, long_lists=True)
    assert torch.allclose(got, ref, rtol=1e-5, atol=2e-5), (got - ref).abs().max().item()
    assert torch.allclose(ops.gather_sum_rows(src, ptr, inv, n), ref, rtol=1e-5, atol=2e-5)
    base = torch.from_numpy(rs.uniform(-1, 1, (n, C)).astype(np.float32)).to(device)
    assert torch.allclose(ops.gather_sum_rows(src, ptr, inv, n, out=base.clone(), long_lists=True), ref + base, rtol=1e-5, atol=2e-5)
    # rows stored in list order (what m3d_lfa_bwd_edge_rows writes): the lists are the rows themselves, no index table
    ordered = torch.zeros_like(src)
    ordered[:int(ptr_h[-1])] = src[inv[:int(ptr_h[-1])].long()]
    assert torch.allclose(ops.gather_sum_rows(ordered, ptr, None, n, long_lists=True), ref, rtol=1e-5, atol=2e-5)
    assert torch.allclose(ops.gather_sum_rows(ordered, ptr, None, n), ref, rtol=1e-5, atol=2e-5)
    # slots alone (no index table): every edge still gets a row of its own inside the list of the point it names
    ptr2, inv2, slot2 = ops.knn_reverse(idx, with_inv=False)
    assert inv2 is None and torch.equal(ptr2, ptr)
    s2 = slot2.cpu().numpy()
    assert np.array_equal(s2 < 0, flat < 0)
    v = flat >= 0
    assert np.array_equal(np.sort(s2[v]), np.arange(int(ptr_h[-1])))
    assert np.all(s2[v] >= ptr_h[flat[v]]) and np.all(s2[v] < ptr_h[flat[v] + 1])


def test_scatter_add_rows_distinct_targets(device):
    """``m3d_scatter_add_rows`` with flags bit 0 (distinct ids: the transpose of decimate()'s subset selection,
    pyg_randla_net.py:234-238) writes what the atomic kernel writes — into zeros and into an existing buffer."""
    from myria3d_amd import ops

    rs = np.random.RandomState(11)
    for n, m, C in ((204800, 51200, 32), (12800, 3200, 256), (100, 100, 8), (1000, 3, 4)):
        idx = torch.from_numpy(rs.permutation(n)[:m].astype(np.int32)).to(device)
        src = torch.from_numpy(rs.uniform(-1, 1, (m, C)).astype(np.float32)).to(device)
        base = torch.from_numpy(rs.uniform(-1, 1, (n, C)).astype(np.float32)).to(device)
        ref = ops.scatter_add_rows(src, idx, n, out=base.clone())
        got = ops.scatter_add_rows(src, idx, n, out=base.clone(), distinct=True)
        assert torch.equal(got, ref)
        assert torch.equal(ops.scatter_add_rows(src, idx, n, out=torch.zeros(n, C, device=device), distinct=True),
                           ops.scatter_add_rows(src, idx, n, out=torch.zeros(n, C, device=device)))


def test_dropout_counter_based_mask(device):
    """``m3d_dropout`` (mlp_classif's Dropout(0.5), pyg_randla_net.py:49-52): kept elements are scaled by 1 / (1 - p), the kept
    fraction is 1 - p, the mask is a function of (seed, device step counter, element) — the same call on dy reproduces the
    forward's mask (the backward pass stores nothing), a bumped counter or another seed draws a new one."""
    from myria3d_amd import ops

    n, c = 204800, 32
    x = torch.rand(n, c, device=device) + 0.5
    counter = torch.zeros(1, dtype=torch.int64, device=device)
    for p in (0.5, 0.1):
        xr = x.clone().requires_grad_(True)
        y = ops.DropoutFn.apply(xr, p, counter, 1234)
        keep = y != 0
        frac = keep.float().mean().item()
        assert abs(frac - (1 - p)) < 2e-3, frac
        assert torch.allclose(y[keep], (x / (1 - p))[keep], rtol=1e-5)
        assert abs(keep.float().mean(0) - (1 - p)).max().item() < 0.01  # no column is favoured
        g = torch.rand_like(x) + 0.5
        y.backward(g)
        assert torch.equal(xr.grad != 0, keep) and torch.allclose(xr.grad[keep], (g / (1 - p))[keep], rtol=1e-5)
        y2 = ops.DropoutFn.apply(x, p, counter, 1234)
        assert torch.equal(y2, y.detach())
        other = ops.DropoutFn.apply(x, p, counter + 1, 1234) != 0
        agree = (other == keep).float().mean().item()
        assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 5e-3, agree  # independent masks
        other = ops.DropoutFn.apply(x, p, counter, 99) != 0
        assert abs((other == keep).float().mean().item() - ((1 - p) ** 2 + p ** 2)) < 5e-3


@pytest.mark.parametrize("M,K,N,k1", [(20000, 64, 32, 0), (4000, 256, 64, 0), (6000, 32, 32, 32), (1000, 64, 32, 0)])
def test_shared_layer_with_fused_dropout_equals_layer_then_dropout(device, M, K, N, k1):
    """Dropout fused into a SharedMLP layer's BatchNorm kernels (``SharedLayerTrainFn(..., drop=...)``: mask applied by
    ``m3d_bn_stats_apply`` on the way out, to the incoming gradient by the column-sum pass and the dz-on-load prologue) gives what the
    layer followed by ``DropoutFn`` gives: output, input gradients and every parameter gradient."""
    from myria3d_amd import ops

    rs = np.random.RandomState(M + K)
    t = lambda *shape: torch.from_numpy(rs.uniform(-1, 1, shape).astype(np.float32)).to(device)
    counter = torch.full((1,), 7, dtype=torch.int64, device=device)
    outs = []
    for fused in (False, True):
        x0 = t(M, K).requires_grad_(True)
        x1 = t(M, k1).requires_grad_(True) if k1 else None
        w = t(N, K + k1).requires_grad_(True)
        b, gamma, beta = t(N).requires_grad_(True), (t(N) + 2).requires_grad_(True), t(N).requires_grad_(True)
        bn = torch.nn.BatchNorm1d(N, eps=1e-6, momentum=0.01).to(device)
        drop = (0.5, counter, 4242)
        y = ops.SharedLayerTrainFn.apply(x0, x1, w, b, gamma, beta, bn, True, None, None, False, None, None,
                                         drop if fused else None)
        if not fused:
            y = ops.DropoutFn.apply(y, *drop)
        g = t(M, N)
        y.backward(g)
        outs.append([y.detach(), x0.grad, x1.grad if k1 else None, w.grad, gamma.grad, beta.grad])
        rs = np.random.RandomState(M + K)  # same draws for the second arm
    for a_, b_, name in zip(outs[0], outs[1], ("y", "dx0", "dx1", "dw", "dgamma", "dbeta")):
        if a_ is None:
            continue
        assert torch.allclose(a_, b_, rtol=2e-5, atol=2e-6 * max(1.0, a_.abs().max().item())), \
            (name, (a_ - b_).abs().max().item())
    assert (outs[1][0] == 0).float().mean().item() > 0.4  # the mask was applied


@pytest.mark.parametrize("M", [1083, 20000])
def test_fused_dropout_follows_the_callers_rows(device, M):
    """``M3DDropout.rows``: a layer that works on a permuted row order (the net's cell-sorted order) with ``rows = perm`` drops the
    same elements of the CALLER's tensor as the layer on the caller's order does — forward and backward; M is not a multiple of
    the 16-row tiles (the dz-on-load prologue must not read ``rows`` for the rows past the end)."""
    from myria3d_amd import ops

    K, N = 64, 32
    rs = np.random.RandomState(M)
    t = lambda *shape: torch.from_numpy(rs.uniform(-1, 1, shape).astype(np.float32)).to(device)
    x, w, g = t(M, K), t(N, K), t(M, N)
    b, gamma, beta = torch.zeros(N, device=device), t(N) + 2, t(N)
    perm = torch.from_numpy(rs.permutation(M).astype(np.int32)).to(device)
    counter = torch.full((1,), 3, dtype=torch.int64, device=device)
    res = []
    for rows in (None, perm):
        xin = (x if rows is None else x[rows.long()]).clone().requires_grad_(True)
        wp = w.clone().requires_grad_(True)
        bn = torch.nn.BatchNorm1d(N, eps=1e-6, momentum=0.01).to(device)
        drop = (0.5, counter, 99) if rows is None else (0.5, counter, 99, rows)
        y = ops.SharedLayerTrainFn.apply(xin, None, wp, b, gamma, beta, bn, True, None, None, False, None, None, drop)
        y.backward(g if rows is None else g[rows.long()])
        res.append((y.detach(), xin.grad, wp.grad))
    (y0, dx0, dw0), (y1, dx1, dw1) = res
    pl = perm.long()
    assert torch.equal(y1 != 0, (y0 != 0)[pl])
    assert torch.allclose(y1, y0[pl], rtol=1e-4, atol=1e-5) and torch.allclose(dx1, dx0[pl], rtol=1e-3, atol=1e-5)
    assert torch.allclose(dw1, dw0, rtol=1e-3, atol=1e-4 * dw0.abs().max().item())


@pytest.mark.parametrize("M,K0,K1,N", [(3200, 256, 256, 512), (12800, 128, 128, 256), (801, 128, 256, 128), (3200, 512, 96, 64),
                                       (5000, 64, 32, 128), (300, 256, 256, 20)])
def test_gemm_pair_equals_two_launches(device, M, K0, K1, N):
    """``m3d_gemm_pair_f32`` (the mlp2 / shortcut Linears of a block as ONE launch on the deep levels, pyg_randla_net.py:172-188):
    outputs and slot-mode statistics are the BITS two ``m3d_gemm_f32`` launches give — the forward pattern with bias and
    statistics, and the input-gradient pattern with one side added into an existing buffer; shapes the pair kernel does not
    take (K <= 64) go through the two-launch fallback inside the same entry point."""
    from myria3d_amd import ops

    rs = np.random.RandomState(M + K0 + N)
    t = lambda *shape: torch.from_numpy(rs.uniform(-1, 1, shape).astype(np.float32)).to(device)
    a, w, b = (t(M, K0), t(M, K1)), (t(N, K0), t(N, K1)), (t(N), t(N))
    pow2 = N >= 4 and (N & (N - 1)) == 0
    st_ref = [torch.zeros((ops.bn_slots(M), 2, N), dtype=torch.float64, device=device) for _ in range(2)] if pow2 else None
    st_got = [torch.zeros_like(x) for x in st_ref] if pow2 else None
    ref = [ops.gemm(a[i], w[i], M, N, a[i].shape[1], bias=b[i], stats=st_ref[i] if pow2 else None, stat_slots=pow2)
           for i in range(2)]
    got = ops.gemm_pair(a, w, M, N, bias=b, stats=st_got)
    for i in range(2):
        assert torch.equal(got[i], ref[i]), (i, (got[i] - ref[i]).abs().max().item())
        if pow2:
            # (fp64 sums of the same fp32 outputs, grouped into slots by a different workgroup count)
            assert torch.allclose(st_got[i].sum(0), st_ref[i].sum(0), rtol=1e-12, atol=1e-9), i
    # input-gradient pattern: dX_i[M, Kin] = dZ_i[M, N] W_i[N, Kin], the second one added into an existing buffer
    Kin = K0
    dz, wd = (t(M, N), t(M, N)), (t(N, Kin), t(N, Kin))
    base = t(M, Kin)
    r0 = ops.linear_dgrad(dz[0], wd[0])
    r1 = ops.linear_dgrad(dz[1], wd[1], acc=base.clone())
    acc = base.clone()
    g0, g1 = ops.gemm_pair(dz, wd, M, Kin, out=(None, acc), accumulate=(False, True), b_cm=True)
    assert g1.data_ptr() == acc.data_ptr()
    assert torch.equal(g0, r0) and torch.equal(g1, r1)


@pytest.mark.parametrize("M,K,N,slots", [(37, 128, 64, False), (800, 512, 512, True), (3200, 256, 512, True),
                                         (204800, 32, 32, True), (51200, 64, 128, False), (20000, 16, 8, True)])
def test_gemm_statistics_of_columns_with_mean_far_from_zero(device, M, K, N, slots):
    """Train-mode BatchNorm statistics from the GEMM epilogue (fp32 partial sums around a per-lane shift, fp64 across lanes)
    for columns whose mean is 1 000 standard deviations away from zero — the case a plain fp32 sum of squares loses: the
    variance recovered from (sum, sum of squares) stays within 1e-4 of the fp64 variance of the same fp32 outputs."""
    from myria3d_amd import ops

    rs = np.random.RandomState(M % 97)
    a = torch.from_numpy(rs.normal(size=(M, K)).astype(np.float32)).to(device)
    w = torch.from_numpy((rs.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)).to(device)
    b = torch.from_numpy(rs.uniform(-1000.0, 1000.0, N).astype(np.float32)).to(device)
    if slots:
        st = ops.stat_slots(N, device, M)
        z = ops.gemm(a, w, M, N, K, bias=b, stats=st, stat_slots=True)
    else:
        st = ops.stat_buffer(M, N, K, device)
        z = ops.gemm(a, w, M, N, K, bias=b, stats=st)
    tot = st.sum(0)  # [2, N]
    zd = z.double()
    mean, var = tot[0] / M, tot[1] / M - (tot[0] / M) ** 2
    ref_mean, ref_var = zd.mean(0), zd.var(0, unbiased=False)
    assert torch.allclose(mean, ref_mean, rtol=1e-9, atol=1e-6), (mean - ref_mean).abs().max().item()
    rel = ((var - ref_var).abs() / ref_var).max().item()
    print(f"[parity] variance of columns with |mean| up to 1000: worst relative error {rel:.2e} (M={M})")
    assert rel < 1e-4, rel

@pytest.mark.parametrize("M,N,K", [(30000, 32, 32), (1001, 6, 32), (777, 32, 9), (1000, 128, 200), (4096, 64, 16),
                                   (3, 512, 768), (12800, 4, 32), (5, 8, 8)])
def test_wgrad_and_dgrad(device, M, N, K):
    """dW = dZ^T X (rows split over workgroups, atomically combined) and dX = dZ W, vs fp64."""
    from myria3d_amd import ops

    rs = np.random.RandomState(M + N + K)
    dz = torch.from_numpy(rs.uniform(-1, 1, (M, N)).astype(np.float32))
    x = torch.from_numpy(rs.uniform(-1, 1, (M, K)).astype(np.float32))
    w = torch.from_numpy(rs.uniform(-1, 1, (N, K)).astype(np.float32))
    dw = ops.linear_wgrad(dz.to(device), x.to(device), K)
    _close("wgrad", dw, dz.double().t() @ x.double(), 1e-5, 3e-6 * np.sqrt(M) + 2e-7 * M)
    sink = torch.ones((N, K), device=device)
    assert ops.linear_wgrad(dz.to(device), x.to(device), K, out=sink) is None
    _close("wgrad.sink", sink, 1.0 + dz.double().t() @ x.double(), 1e-5, 3e-6 * np.sqrt(M) + 2e-7 * M)
    dx = ops.linear_dgrad(dz.to(device), w.to(device))
    _close("dgrad", dx, dz.double() @ w.double(), 1e-5, 2e-6 * N)


def test_gemm_gather_concat_and_transposes(device):
    from myria3d_amd import ops

    rs = np.random.RandomState(7)
    n_c, M, k0, k1, N = 90, 333, 128, 32, 32
    xc = torch.from_numpy(rs.uniform(-1, 1, (n_c, k0)).astype(np.float32))
    xs = torch.from_numpy(rs.uniform(-1, 1, (M, k1)).astype(np.float32))
    rows = torch.from_numpy(rs.randint(0, n_c, (M,)).astype(np.int32))
    w = torch.from_numpy(rs.uniform(-1, 1, (N, k0 + k1)).astype(np.float32))
    ref = torch.cat([xc[rows.long()], xs], 1).double() @ w.double().t()
    got = ops.gemm(xc.to(device), w.to(device), M, N, k0, rows=rows.to(device), a1=xs.to(device), k1=k1)
    _close("gemm.gather_concat", got, ref, 1e-5, 4e-4)
    # dgrad: dX = dZ W
    dz = torch.from_numpy(rs.uniform(-1, 1, (M, N)).astype(np.float32))
    _close("gemm.dgrad", ops.linear_dgrad(dz.to(device), w.to(device)), dz.double() @ w.double(), 1e-5, 1e-4)
    # wgrad: dW = dZ^T [X0[rows] | X1]   (split-K + atomics)
    dw = ops.linear_wgrad(dz.to(device), xc.to(device), k0, rows.to(device), xs.to(device), k1)
    _close("gemm.wgrad", dw, dz.double().t() @ torch.cat([xc[rows.long()], xs], 1).double(), 1e-5, 2e-3)
    big = torch.from_numpy(rs.uniform(-1, 1, (20000, 16)).astype(np.float32))
    dzb = torch.from_numpy(rs.uniform(-1, 1, (20000, 8)).astype(np.float32))
    _close("gemm.wgrad_long", ops.linear_wgrad(dzb.to(device), big.to(device), 16), dzb.double().t() @ big.double(),
           1e-5, 5e-3)
    _close("colsum", ops.colsum(dzb.to(device)), dzb.double().sum(0), 1e-5, 5e-3)


@pytest.mark.parametrize("M,N,ld", [(1083, 32, 32), (204800, 32, 32), (5000, 64, 96), (3001, 6, 6), (777, 1024, 1024),
                                     (100, 2048, 2048), (1, 8, 8), (40000, 4, 4)])
def test_colsum_row_shapes(device, M, N, ld):
    """Bias gradients (``m3d_colsum_f32``): the float4 kernel (N / 4 a power of two <= 256, ld a multiple of 4 — also a column
    slice of a wider matrix), the dword kernel for everything else (N = 6: fc_classif), row counts that end inside a trip of
    eight rows, and ``out=`` as a gradient sink that is ADDED to."""
    from myria3d_amd import ops

    rs = np.random.RandomState(M + N)
    full = torch.from_numpy(rs.uniform(-1, 1, (M, ld)).astype(np.float32)).to(device)
    x = full[:, :N]
    ref = x.double().sum(0)
    _close("colsum", ops.colsum(x), ref, 1e-5, 2e-4 * max(1.0, M ** 0.5))
    sink = torch.full((N,), 3.0, device=device)
    assert ops.colsum(x, out=sink) is None
    _close("colsum.sink", sink, ref + 3.0, 1e-5, 2e-4 * max(1.0, M ** 0.5))


# ----------------------------------------------------------------------------------------------- SharedMLP layer (train)
def _cpu_layer(x, w, b, gamma, beta, act):
    lin = torch.nn.Linear(w.shape[1], w.shape[0])
    bn = torch.nn.BatchNorm1d(w.shape[0], eps=1e-6, momentum=0.01)
    with torch.no_grad():
        lin.weight.copy_(w), lin.bias.copy_(b), bn.weight.copy_(gamma), bn.bias.copy_(beta)
    lin, bn = lin.double(), bn.double()
    y = bn(lin(x))
    return (torch.nn.functional.leaky_relu(y, 0.2) if act else y), lin, bn


@pytest.mark.parametrize("M,K,N,act", [(1000, 32, 32, True), (517, 64, 128, False), (3, 512, 512, True),
                                        (20011, 64, 64, True), (3333, 128, 16, True), (801, 16, 256, False),
                                        (1000, 24, 40, True)])
def test_shared_layer_train_fwd_bwd(device, M, K, N, act):
    from myria3d_amd import ops

    rs = np.random.RandomState(M)
    x = torch.from_numpy(rs.uniform(-1, 1, (M, K))).double().requires_grad_(True)
    w = torch.from_numpy(rs.uniform(-1, 1, (N, K)) / np.sqrt(K)).float()
    b = torch.from_numpy(rs.uniform(-1, 1, (N,))).float()
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, (N,))).float()
    beta = torch.from_numpy(rs.uniform(-0.5, 0.5, (N,))).float()
    gy = torch.from_numpy(rs.uniform(-1, 1, (M, N))).double()
    y_ref, lin, bn_ref = _cpu_layer(x, w, b, gamma, beta, act)
    y_ref.backward(gy)

    bn = torch.nn.BatchNorm1d(N, eps=1e-6, momentum=0.01).to(device)
    xg = x.detach().float().to(device).requires_grad_(True)
    wg, bg = w.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
    with torch.no_grad():
        bn.weight.copy_(gamma), bn.bias.copy_(beta)
    y = ops.SharedLayerTrainFn.apply(xg, None, wg, bg, bn.weight, bn.bias, bn, act, None)
    y.backward(gy.float().to(device))
    s = 1.0 if M > 10 else 30.0  # tiny batches: 1/std amplification
    _close("layer.y", y, y_ref, 1e-4 * s, 1e-5 * s)
    _close("layer.running_mean", bn.running_mean, bn_ref.running_mean, 1e-5, 1e-6)
    _close("layer.running_var", bn.running_var, bn_ref.running_var, 1e-5, 1e-6)
    assert int(bn.num_batches_tracked) == 1
    _close("layer.dx", xg.grad, x.grad, 1e-3 * s, 1e-5 * s)
    _close("layer.dW", wg.grad, lin.weight.grad, 1e-3 * s, 1e-4 * s)
    _close("layer.dgamma", bn.weight.grad, bn_ref.weight.grad, 1e-3 * s, 1e-4 * s)
    _close("layer.dbeta", bn.bias.grad, bn_ref.bias.grad, 1e-3 * s, 1e-4 * s)
    _close("layer.dbias", bg.grad, lin.bias.grad, 0, 1e-3)  # analytically zero


@pytest.mark.parametrize("M,K,N,act,bf16", [(20011, 8, 8, True, False), (5000, 32, 16, False, False),
                                             (4099, 64, 32, True, False), (12800, 32, 64, True, False),
                                             (3200, 256, 128, True, False), (801, 96, 512, True, False),
                                             (3200, 256, 128, True, True), (37, 512, 256, False, True)])
def test_bn_dgrad_fused_matches_the_two_pass_backward(device, M, K, N, act, bf16):
    """m3d_bn_dgrad_f32 (BatchNorm backward as the A-prologue of the dgrad GEMM) against m3d_bn_bwd + m3d_gemm_f32: same
    arithmetic per element, so dz / dgamma / dbeta agree to rounding of the fp64 slot sums and dx to fp32 GEMM order."""
    from myria3d_amd import ops

    rs = np.random.RandomState(M + N)
    x = torch.from_numpy(rs.uniform(-1, 1, (M, K)).astype(np.float32)).to(device)
    w = torch.from_numpy((rs.uniform(-1, 1, (N, K)) / np.sqrt(K)).astype(np.float32)).to(device)
    b = torch.zeros(N, device=device)
    gy = torch.from_numpy(rs.uniform(-1, 1, (M, N)).astype(np.float32)).to(device)
    res = {}
    for fused in (True, False):
        ops.FUSE_BN_DGRAD = fused
        try:
            bn = torch.nn.BatchNorm1d(N, eps=1e-6, momentum=0.01).to(device)
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, N)), bn.bias.copy_(torch.linspace(-0.3, 0.3, N))
            xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = ops.SharedLayerTrainFn.apply(xg, None, wg, b, bn.weight, bn.bias, bn, act, None, None, bf16)
            y.backward(gy)
            res[fused] = (xg.grad.clone(), wg.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
        finally:
            ops.FUSE_BN_DGRAD = True
    tol = 3e-2 if bf16 else 2e-5
    for name, a, r in zip(("dx", "dW", "dgamma", "dbeta"), res[True], res[False]):
        _relclose(f"bn_dgrad.{name}", a, r, tol)


def test_residual_tail_train(device):
    from myria3d_amd import ops

    rs = np.random.RandomState(11)
    M, K2, Ks, N = 700, 16, 32, 32
    x2 = torch.from_numpy(rs.uniform(-1, 1, (M, K2))).double().requires_grad_(True)
    xs = torch.from_numpy(rs.uniform(-1, 1, (M, Ks))).double().requires_grad_(True)
    prm = [torch.from_numpy(rs.uniform(-1, 1, s)).float() for s in ((N, K2), (N,), (N,), (N,), (N, Ks), (N,), (N,), (N,))]
    prm[2] = prm[2] * 0.5 + 1.0
    prm[6] = prm[6] * 0.5 + 1.0
    y2, lin2, bn2r = _cpu_layer(x2, prm[0], prm[1], prm[2], prm[3], False)
    ys, lins, bnsr = _cpu_layer(xs, prm[4], prm[5], prm[6], prm[7], False)
    ref = torch.nn.functional.leaky_relu(y2 + ys, 0.2)
    gy = torch.from_numpy(rs.uniform(-1, 1, (M, N))).double()
    ref.backward(gy)
    bn2 = torch.nn.BatchNorm1d(N, eps=1e-6, momentum=0.01).to(device)
    bns = torch.nn.BatchNorm1d(N, eps=1e-6, momentum=0.01).to(device)
    with torch.no_grad():
        bn2.weight.copy_(prm[2]), bn2.bias.copy_(prm[3]), bns.weight.copy_(prm[6]), bns.bias.copy_(prm[7])
    g = lambda t: t.detach().float().to(device).requires_grad_(True)
    x2g, xsg, w2, b2, ws_, bs_ = g(x2), g(xs), g(prm[0]), g(prm[1]), g(prm[4]), g(prm[5])
    y = ops.ResidualTailTrainFn.apply(x2g, w2, b2, bn2.weight, bn2.bias, bn2, xsg, ws_, bs_, bns.weight, bns.bias, bns)
    y.backward(gy.float().to(device))
    _close("tail.y", y, ref, 1e-4, 1e-5)
    _close("tail.dx2", x2g.grad, x2.grad, 1e-3, 1e-5)
    _close("tail.dxs", xsg.grad, xs.grad, 1e-3, 1e-5)
    _close("tail.dW2", w2.grad, lin2.weight.grad, 1e-3, 1e-4)
    _close("tail.dWs", ws_.grad, lins.weight.grad, 1e-3, 1e-4)
    _close("tail.dgamma2", bn2.weight.grad, bn2r.weight.grad, 1e-3, 1e-4)
    _close("tail.dbetas", bns.bias.grad, bnsr.bias.grad, 1e-3, 1e-4)


def test_attention_weight_pack_kernel_matches_host_packing(device):
    from myria3d_amd import ops

    for ch in (8, 16, 64, 256):
        w = torch.randn(ch, ch, device=device)
        wp, wpt = ops.pack_attention_weights(w, True)
        assert torch.equal(wp, ops.pack_attention_weight(w).reshape(-1))
        assert torch.equal(wpt, ops.pack_attention_weight(w.t()).reshape(-1))


# ----------------------------------------------------------------------------------------------- rows / decimation
def test_gather_scatter_decimation(device):
    from myria3d_amd import ops

    rs = np.random.RandomState(5)
    src = torch.from_numpy(rs.uniform(-1, 1, (1000, 32)).astype(np.float32))
    idx = torch.from_numpy(rs.randint(0, 1000, (4000,)).astype(np.int32))
    got = ops.gather_rows(src.to(device), idx.to(device))
    assert torch.equal(got.cpu(), src[idx.long()])
    src3 = src[:, :3].contiguous()
    assert torch.equal(ops.gather_rows(src3.to(device), idx.to(device)).cpu(), src3[idx.long()])
    sc = ops.scatter_add_rows(got, idx.to(device), 1000)
    ref = torch.zeros(1000, 32, dtype=torch.float64).index_add_(0, idx.long(), src[idx.long()].double())
    _close("scatter_add", sc, ref, 1e-6, 1e-5)
    p4 = ops.pad_pos(src3.to(device)).cpu()
    assert torch.equal(p4[:, :3], src3) and bool((p4[:, 3] == 0).all())
    # decimation: per cloud, m distinct in-range indices; different seeds/levels give different subsets
    sizes = [1000, 37, 4, 1, 12800]
    ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64)
    new = [max(1, n // 4) for n in sizes]
    ptr_out = torch.tensor([0] + list(np.cumsum(new)), dtype=torch.int64)
    seed = torch.tensor([12345], dtype=torch.int64, device=device)
    a = ops.decimation_indices(ptr.to(device), ptr_out.to(device), int(ptr_out[-1]), seed, 0).cpu().long()
    b = ops.decimation_indices(ptr.to(device), ptr_out.to(device), int(ptr_out[-1]), seed, 1).cpu().long()
    for c in range(len(sizes)):
        seg = a[ptr_out[c]:ptr_out[c + 1]]
        assert seg.numel() == new[c] and seg.unique().numel() == new[c]
        assert bool((seg >= ptr[c]).all()) and bool((seg < ptr[c + 1]).all())
    assert not torch.equal(a, b)
    # the keyed permutation is a bijection: asking for ALL slots returns every point exactly once
    full = ops.decimation_indices(ptr.to(device), ptr.to(device), int(ptr[-1]), seed, 0).cpu().long()
    assert torch.equal(full.sort().values, torch.arange(int(ptr[-1])))
    # roughly uniform: mean of selected local ranks ~ n/2
    seg = (a[ptr_out[4]:ptr_out[5]] - ptr[4]).double()
    assert abs(seg.mean().item() / 12800 - 0.5) < 0.03


def test_decimation_draws_look_uniform(device):
    """The keyed Feistel permutation behind m3d_decimation_indices (rows.hip) stands in for torch.randperm
    (pyg_randla_net.py:221): every point must survive with probability m / n and pairs of points with probability
    m (m - 1) / (n (n - 1)), independently from cloud to cloud.  20 000 clouds of 40 points, 10 survivors each:
    chi-square of the per-point survival counts (39 degrees of freedom) and z-scores of all 780 pair counts; 2 000
    clouds of 1 000 points for a longer permutation."""
    from myria3d_amd import ops

    def draw(B, n, seed_val):
        m = n // 4
        ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=device)
        ptr_out = torch.arange(0, (B + 1) * m, m, dtype=torch.int64, device=device)
        seed = torch.tensor([seed_val], dtype=torch.int64, device=device)
        idx = ops.decimation_indices(ptr, ptr_out, B * m, seed, 0).cpu().long().view(B, m)
        local = idx - (torch.arange(B) * n)[:, None]
        assert bool((local >= 0).all()) and bool((local < n).all())
        sel = torch.zeros(B, n, dtype=torch.bool)
        sel[torch.arange(B)[:, None], local] = True
        assert bool((sel.sum(1) == m).all())  # distinct survivors in every cloud
        return sel.double(), m

    B, n = 20000, 40
    sel, m = draw(B, n, 0x123456789ABCDEF)
    p1 = m / n
    cnt = sel.sum(0)
    chi2 = (((cnt - B * p1) ** 2) / (B * p1 * (1 - p1))).sum().item()
    p2 = p1 * (m - 1) / (n - 1)
    co = sel.t() @ sel
    iu = torch.triu_indices(n, n, 1)
    z = (co[iu[0], iu[1]] - B * p2) / (B * p2 * (1 - p2)) ** 0.5
    print(f"[stats] decimation: chi2(39 dof) = {chi2:.1f}, pair z: max |z| = {z.abs().max().item():.2f}, "
          f"mean z^2 = {(z ** 2).mean().item():.2f}")
    assert chi2 < 90.0            # P(chi2_39 > 90) ~ 1e-5
    assert z.abs().max().item() < 5.5 and (z ** 2).mean().item() < 1.5
    sel, m = draw(2000, 1000, 77)
    cnt = sel.sum(0)
    chi2 = (((cnt - 2000 * 0.25) ** 2) / (2000 * 0.25 * 0.75)).sum().item()
    print(f"[stats] decimation, n = 1000: chi2(999 dof) = {chi2:.1f}")
    assert 800.0 < chi2 < 1250.0  # mean 999, sd 44.7


@pytest.mark.parametrize("M,K,N,act", [(20000, 16, 32, True), (5003, 64, 128, True), (1083, 32, 64, False), (204800, 64, 32, True),
                                        (777, 8, 16, True)])
def test_gemm_with_batchnorm_applied_on_load(device, M, K, N, act):
    """Round 5: ``m3d_gemm_bn_on_load_f32`` (the SharedMLP layer behind another one: BatchNorm + LeakyReLU of the layer in
    front applied to the A fragments as they are loaded) against the two launches it replaces, ``m3d_bn_stats_apply`` +
    ``m3d_gemm_f32`` — product, its slot statistics, the stored activation, the four per-column vectors of the backward pass
    and the running statistics."""
    from myria3d_amd import ops

    rs = np.random.RandomState(M % 97 + K)
    z = torch.from_numpy((rs.normal(0.3, 1.5, (M, K)) * rs.uniform(0.2, 3.0, (1, K))).astype(np.float32)).to(device)
    w = torch.from_numpy(rs.normal(0, K ** -0.5, (N, K)).astype(np.float32)).to(device)
    b = torch.from_numpy(rs.normal(0, 0.1, (N,)).astype(np.float32)).to(device)

    def bn_module():
        bn = torch.nn.BatchNorm1d(K, eps=1e-6, momentum=0.01).to(device)
        with torch.no_grad():
            bn.weight.copy_(torch.from_numpy(rs.uniform(0.5, 1.5, K).astype(np.float32)))
            bn.bias.copy_(torch.from_numpy(rs.normal(0, 0.2, K).astype(np.float32)))
            bn.running_mean.fill_(0.25)
            bn.running_var.fill_(2.0)
        return bn

    state = rs.get_state()
    bn_a = bn_module()
    rs.set_state(state)
    bn_b = bn_module()
    ops.arena.stop()
    # the statistics of the layer in front: column sums / sums of squares of z in slot mode (here: from torch, in slot 0)
    slots = torch.zeros((ops.bn_slots(M), 2, K), dtype=torch.float64, device=device)
    slots[0, 0] = z.double().sum(0)
    slots[0, 1] = (z.double() ** 2).sum(0)
    # reference: the two launches
    y_ref, vec_ref = ops.bn_stats_apply(slots, M, bn_a, z, act)
    st_ref = torch.zeros((ops.bn_slots(M), 2, N), dtype=torch.float64, device=device)
    c_ref = ops.gemm(y_ref, w, M, N, K, bias=b, stats=st_ref, stat_slots=True)
    # fused
    pv = torch.empty((4, K), dtype=torch.float32, device=device).unbind(0)
    pend = ops.PendingBN(z, slots, M, bn_b, act, torch.empty_like(z), pv)
    st = torch.zeros_like(st_ref)
    c = ops.gemm_bn_on_load(pend, w, M, N, b, st)
    if K % 4 or K > 64:
        assert c is None
        return
    assert c is not None and pend.done
    _close("bn_on_load.C", c, c_ref, 1e-5, 1e-5)
    assert torch.equal(pend.y, y_ref), "the stored activation: the same arithmetic as m3d_bn_stats_apply"
    for name, a_, b_ in zip(("scale", "shift", "mean", "invstd"), pend.vecs, vec_ref):
        assert torch.equal(a_, b_), name
    assert torch.equal(bn_b.running_mean, bn_a.running_mean) and torch.equal(bn_b.running_var, bn_a.running_var)
    assert int(bn_b.num_batches_tracked) == int(bn_a.num_batches_tracked) == 1
    tot, tot_ref = st.sum(0), st_ref.sum(0)
    assert torch.allclose(tot, tot_ref, rtol=1e-9, atol=1e-6 * float(tot_ref.abs().max()))


# ----------------------------------------------------------------------------------------------- LFA
def _lfa_setup(ch, sizes, k, seed):
    from oracle.randla_oracle import LocalFeatureAggregation, dense_to_edge_index, knn_exact

    x, pos, _, ptr = rand_batch(sizes, num_features=ch // 2, seed=seed)
    x = x * 2 - 1
    lfa = LocalFeatureAggregation(ch)
    fill_params_deterministic(lfa, seed)
    idx, _ = knn_exact(pos, ptr.tolist(), pos, ptr.tolist(), k)
    return x, pos, ptr, lfa, idx, dense_to_edge_index(idx)


@pytest.mark.parametrize("ch", [8, 16, 32, 64, 128, 256])
@pytest.mark.parametrize("k", [16, 32])
def test_lfa_eval_forward(device, ch, k):
    from myria3d_amd import ops

    x, pos, ptr, lfa, idx, ei = _lfa_setup(ch, [150, 9, 77], k, seed=ch + k)
    lfa.eval()
    with torch.no_grad():
        ref = lfa.aggregate(ei, x, pos)
    enc_lin, enc_bn = lfa.mlp_encoder.lins[0].to(device), lfa.mlp_encoder.norms[0].module.to(device)
    w_att = lfa.mlp_attention.lins[0].weight.to(device)
    wf, bf, _, _ = ops.lfa_enc_fold(enc_lin, enc_bn, None, 0)
    pos4 = ops.pad_pos(pos.to(device))
    idx32 = idx.to(torch.int32).to(device)
    got = ops.lfa_forward(x.to(device), pos4, idx32, wf, bf, w_att)
    _close(f"lfa_fwd(ch={ch},k={k})", got, ref, 2e-5, 2e-5)
    got_u = ops.lfa_forward_unfused(x.to(device), pos4, idx32, wf, bf, w_att)
    _close(f"lfa_unfused(ch={ch},k={k})", got_u, ref, 2e-5, 2e-5)


@pytest.mark.parametrize("ch", [8, 16, 32, 64, 128, 256])
@pytest.mark.parametrize("k", [16, 32])
@pytest.mark.parametrize("sizes", [[151, 41, 77], [16 * 40 + 32]])
def test_lfa_forward_full_neighbourhood_kernel(device, ch, k, sizes):
    """Round 5: the mask-free forward kernel (``M3D_LFA_FULL``: complete neighbourhoods promised; two centres per MFMA
    tile at ch = 8, 32-bit offsets, exp2-fma softmax) against the oracle AND against the general kernel on the same
    inputs.  An odd point count (the last centre pair of the ch = 8 tiles is half empty, the last workgroup partly
    empty) and one that is a multiple of every tile."""
    from myria3d_amd import ops

    x, pos, ptr, lfa, idx, ei = _lfa_setup(ch, sizes, k, seed=ch + k + len(sizes))
    assert min(sizes) >= k and int(idx.min()) >= 0
    lfa.eval()
    with torch.no_grad():
        ref = lfa.aggregate(ei, x, pos)
    enc_lin, enc_bn = lfa.mlp_encoder.lins[0].to(device), lfa.mlp_encoder.norms[0].module.to(device)
    w_att = lfa.mlp_attention.lins[0].weight.to(device)
    wf, bf, _, _ = ops.lfa_enc_fold(enc_lin, enc_bn, None, 0)
    pos4 = ops.pad_pos(pos.to(device))
    idx32 = idx.to(torch.int32).to(device)
    got = ops.lfa_forward(x.to(device), pos4, idx32, wf, bf, w_att, full=True)
    gen = ops.lfa_forward(x.to(device), pos4, idx32, wf, bf, w_att, full=False)
    _close(f"lfa_fwd_full(ch={ch},k={k})", got, ref, 2e-5, 2e-5)
    _close(f"lfa_fwd_full vs general (ch={ch},k={k})", got, gen, 1e-5, 1e-5)


@pytest.mark.parametrize("ch,k", [(8, 16), (16, 16), (32, 16), (64, 16), (128, 16), (256, 16), (8, 32), (16, 32), (64, 32),
                                  (256, 32)])
def test_lfa_train_full_neighbourhoods(device, ch, k):
    """Round 5: forward + fused backward on the mask-free kernels (every cloud has >= K points, so ``LFATrainFn`` passes
    ``M3D_LFA_FULL`` / flags bit 3) against the fp64 oracle, at a point count that leaves the last group partly empty."""
    _lfa_train_parity(device, ch, k, [203, 41, 91], seed=ch + 1, fused=True)


@pytest.mark.parametrize("ch,k,n", [(64, 16, 335), (128, 16, 335), (256, 16, 335), (64, 32, 400), (256, 32, 400), (64, 16, 17000),
                                    (128, 16, 8500)])
def test_lfa_split_bf16_products_meet_the_fp32_tolerances(device, ch, k, n):
    """Round 5 experiment: the three attention GEMMs of the LFA kernels as SPLIT-bf16 products — operands
    x = hi + lo (two bf16 values, 16 mantissa bits), hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32 accumulate — against
    the fp64 oracle at the UNCHANGED tolerances of the fp32 kernels (forward 1e-4, dx 1e-3, parameter gradients 1e-3 / 2e-3),
    small sizes and sizes that keep every persistent workgroup in its loop."""
    third = n // 3
    _lfa_train_parity(device, ch, k, [third, third + 1, n - 2 * third - 1], seed=ch + 2, fused=True, big=n > 1000, mode=2)


def _relclose(name, got, ref, rel):
    """Reduced quantities (sums over all edges): relative L2 error of the whole tensor."""
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    den = ref.norm().item()
    err = (got - ref).norm().item()
    print(f"[parity] {name}: rel_l2_err={err / max(den, 1e-30):.3e} ref_norm={den:.3e}")
    assert err <= rel * den + 1e-7, f"{name}: relative L2 error {err / max(den, 1e-30):.3e} > {rel}"


def _lfa_train_parity(device, ch, k, sizes, seed, fused=True, big=False, mode=0, edge_rows=False):
    """m3d_lfa_fwd + m3d_lfa_bwd (train mode: encoder BatchNorm on batch statistics) vs the fp64 oracle
    (LocalFeatureAggregation.aggregate + autograd).  ``big``: kNN table through cKDTree (any valid table serves an
    op-level check), reduced gradients compared by relative L2 norm."""
    from myria3d_amd import ops
    from oracle.randla_oracle import LocalFeatureAggregation, dense_to_edge_index, knn_exact, knn_kdtree

    ops.LFATrainFn.force_unfused_backward = not fused
    x, pos, _, ptr = rand_batch(sizes, num_features=ch // 2, seed=seed)
    x = x * 2 - 1
    lfa = LocalFeatureAggregation(ch)
    fill_params_deterministic(lfa, seed)
    idx, _ = (knn_kdtree if big else knn_exact)(pos, ptr.tolist(), pos, ptr.tolist(), k)
    ei = dense_to_edge_index(idx)
    lfa = lfa.double().train()
    xr = x.double().requires_grad_(True)
    ref = lfa.aggregate(ei, xr, pos.double())
    gy = torch.from_numpy(np.random.RandomState(ch).uniform(-1, 1, tuple(ref.shape)))
    ref.backward(gy)
    enc_lin_r, enc_bn_r = lfa.mlp_encoder.lins[0], lfa.mlp_encoder.norms[0].module

    g = LocalFeatureAggregation(ch)
    fill_params_deterministic(g, seed)
    g = g.to(device).train()
    enc_lin, enc_bn = g.mlp_encoder.lins[0], g.mlp_encoder.norms[0].module
    w_att = g.mlp_attention.lins[0].weight
    pos4 = ops.pad_pos(pos.to(device))
    idx32 = idx.to(torch.int32).to(device)
    num_edges = sum(n * min(k, n) for n in sizes)
    mom = ops.lfa_moments(pos4, idx32)
    xg = x.to(device).requires_grad_(True)
    rev = None
    if edge_rows:  # the input gradient stored per edge and summed over every point's reverse neighbour list (no atomics)
        assert ops.lib().m3d_lfa_bwd_edge_rows_ok(idx32.shape[0], k, ch, ops.LRELU_SLOPE) == 1 and num_edges == idx32.numel()
        rev = ops.knn_reverse(idx32)
    out = ops.LFATrainFn.apply(xg, pos4, idx32, mom, num_edges, enc_lin.weight, enc_lin.bias, enc_bn.weight,
                               enc_bn.bias, enc_lin, enc_bn, w_att, None, mode, None, rev)
    out.backward(gy.float().to(device))
    sc = max(1.0, gy.abs().max().item())
    checks = [
        (_close, f"lfa_train.out(ch={ch})", out, ref, 1e-4, 1e-4),
        (_close, "lfa_train.running_mean", enc_bn.running_mean, enc_bn_r.running_mean, 1e-4, 1e-5),
        (_close, "lfa_train.running_var", enc_bn.running_var, enc_bn_r.running_var, 1e-4, 1e-5),
        (_close, "lfa_train.dx", xg.grad, xr.grad, 1e-3, 1e-4 * sc),
    ]
    if big:
        checks += [
            (_relclose, "lfa_train.dW_att", w_att.grad, lfa.mlp_attention.lins[0].weight.grad, 1e-3),
            (_relclose, "lfa_train.dW_enc", enc_lin.weight.grad, enc_lin_r.weight.grad, 2e-3),
            (_relclose, "lfa_train.dgamma_enc", enc_bn.weight.grad, enc_bn_r.weight.grad, 2e-3),
            (_relclose, "lfa_train.dbeta_enc", enc_bn.bias.grad, enc_bn_r.bias.grad, 2e-3),
        ]
    else:
        checks += [
            (_close, "lfa_train.dW_att", w_att.grad, lfa.mlp_attention.lins[0].weight.grad, 1e-3, 1e-3),
            (_close, "lfa_train.dW_enc", enc_lin.weight.grad, enc_lin_r.weight.grad, 2e-3, 2e-3),
            (_close, "lfa_train.dgamma_enc", enc_bn.weight.grad, enc_bn_r.weight.grad, 2e-3, 2e-3),
            (_close, "lfa_train.dbeta_enc", enc_bn.bias.grad, enc_bn_r.bias.grad, 2e-3, 2e-3),
            (_close, "lfa_train.db_enc", enc_lin.bias.grad, enc_lin_r.bias.grad, 0, 2e-3),
        ]
    failures = []
    for fn, *c in checks:
        try:
            fn(*c)
        except AssertionError as e:
            failures.append(str(e))
    ops.LFATrainFn.force_unfused_backward = False
    assert not failures, failures


@pytest.mark.parametrize("ch,k,fused", [(8, 16, True), (16, 16, True), (32, 16, True), (64, 16, True), (128, 16, True),
                                        (256, 16, True), (16, 32, True), (128, 32, True), (8, 16, False),
                                        (64, 16, False), (32, 40, True)])
def test_lfa_train_forward_backward(device, ch, k, fused):
    """fused=True: m3d_lfa_bwd; fused=False: the materialising fallback (also what K > 32 uses)."""
    _lfa_train_parity(device, ch, k, [200, 11, 90], seed=ch, fused=fused)


# sizes at which EVERY persistent workgroup of lfa_bwd_kernel walks >= 4 groups (its grid is capped at
# 1024 / 1024 / 1024 / 512 / 256 workgroups of 8 / 4 / 4 / 4 / 4 centres for ch <= 16 / 32 / 64 / 128 / 256 at K = 16,
# half as many centres per group at K = 32), so the grid-stride loop, the register prefetch of the NEXT group and the
# double-buffered neighbour ids (PIPE, ch <= 64) are all compared with the oracle, not only the first trip
_PERSISTENT_CASES = [(8, 16, 100000), (16, 16, 50000), (32, 16, 17000), (64, 16, 17000), (128, 16, 8500),
                     (256, 16, 4300), (16, 32, 17000), (64, 32, 8500), (256, 32, 2200)]


@pytest.mark.parametrize("ch,k,n", _PERSISTENT_CASES)
def test_lfa_backward_persistent_loop(device, ch, k, n):
    """The launch shape the bench times (many groups per workgroup) against the fp64 oracle."""
    from myria3d_amd import _lib

    kp = 16 if k <= 16 else 32
    chp = max(ch, 16)
    rows = 128 if chp == 16 else 64
    cap = {16: 1024, 32: 1024, 64: 1024, 128: 512, 256: 256}[chp]
    if chp == 16 and k == 16:  # round 5: the wave-autonomous kernel (complete neighbourhoods): 768 workgroups x 4 waves of
        rows, cap = 8 * ch, 768  # 8 / 4 centres, ids two trips and rows one trip ahead of the arithmetic
    groups = -(-n // (rows // kp))
    assert groups >= 4 * cap, "test sizes must keep every workgroup in its loop for >= 4 trips"
    # the workspace query reports the grid the launcher will use: parts = grid * kspl3
    assert _lib.lib().m3d_lfa_bwd_workspace_bytes(n, k, ch) > 0
    third = n // 3
    _lfa_train_parity(device, ch, k, [third, third + 7, n - 2 * third - 7], seed=ch + k, big=True)


@pytest.mark.parametrize("ch,sizes,big", [(8, [200, 17, 90], False), (16, [200, 17, 90], False), (16, [16], False),
                                          (8, [33000, 33007, 34000], True), (16, [16600, 16607, 16800], True)])
@pytest.mark.parametrize("slots", [True, False])
def test_lfa_backward_edge_rows_and_reverse_lists(device, ch, sizes, big, slots):
    """Round 5: ``m3d_lfa_bwd`` flags bit 5 — the 8 / 16-channel layers store their input gradient per EDGE ([n K, D] rows,
    no atomics) and ``m3d_gather_sum_rows`` sums every point's reverse neighbour list (CSR inverse of the K-NN table) —
    against the fp64 oracle, at the tolerances of the atomic scatter."""
    from myria3d_amd import ops

    keep = ops.USE_LFA_EDGE_SLOTS
    ops.USE_LFA_EDGE_SLOTS = slots  # rows in reverse-list order (contiguous per point) / in edge order (gather through inv)
    try:
        _lfa_train_parity(device, ch, 16, sizes, seed=3 * ch, big=big, edge_rows=True)
    finally:
        ops.USE_LFA_EDGE_SLOTS = keep


def test_lfa_backward_edge_rows_are_declined_where_no_kernel_stores_them(device):
    from myria3d_amd import _lib, ops

    lib = _lib.lib()
    assert lib.m3d_lfa_bwd_edge_rows_ok(1000, 16, 8, 0.2) == 1 and lib.m3d_lfa_bwd_edge_rows_ok(1000, 16, 16, 0.2) == 1
    assert lib.m3d_lfa_bwd_edge_rows_ok(1000, 16, 32, 0.2) == 0   # four-wave tile kernel: atomics
    assert lib.m3d_lfa_bwd_edge_rows_ok(1000, 32, 16, 0.2) == 0   # K = 32
    assert lib.m3d_lfa_bwd_edge_rows_ok(1000, 16, 16, 1.5) == 0   # LeakyReLU is written max(v, slope v)
    assert lib.m3d_lfa_bwd_edge_rows_ok(1 << 27, 16, 16, 0.2) == 0  # 32-bit byte offsets
    n, ch, K = 64, 32, 16
    z = lambda *s: torch.zeros(*s, device=device)
    idx = torch.zeros(n, K, dtype=torch.int32, device=device)
    ws = torch.empty(lib.m3d_lfa_bwd_workspace_bytes(n, K, ch), dtype=torch.uint8, device=device)
    p = lambda t: t.data_ptr()
    rc = lib.m3d_lfa_bwd(p(z(n, ch // 2)), p(z(n, 4)), p(idx), n, K, ch, p(z(ch // 2, 10)), p(z(ch // 2)), p(z(ch * ch)),
                         p(z(ch * ch)), 0.2, p(z(n, ch)), p(z(n * K, ch // 2)), p(z(ch, ch)), 2 | 8 | 32,
                         p(torch.zeros(11 * ch // 2, dtype=torch.float64, device=device)), p(ws), None)
    assert rc == -2  # M3D_ERR_UNSUPPORTED


@pytest.mark.parametrize("ch,k,n", [(32, 16, 3000), (32, 32, 1200), (64, 16, 3000), (128, 16, 2500), (256, 16, 1500), (64, 32, 1500), (256, 32, 900),
                                    (64, 16, 17000)])
def test_lfa_bf16_matrix_core_variant(device, ch, k, n):
    """BASELINE config 2's "bf16": the attention GEMMs of the LFA kernels on bf16 matrix cores (fp32 accumulate, fp32
    everything else) vs the fp64 oracle.  Tolerances follow from bf16's 8-bit mantissa (relative 2^-9 per operand):
    forward 2e-2 absolute on O(1) outputs, gradients 3e-2 relative L2."""
    from myria3d_amd import ops
    from oracle.randla_oracle import LocalFeatureAggregation, dense_to_edge_index, knn_kdtree

    third = n // 3
    sizes = [third, third + 5, n - 2 * third - 5]
    x, pos, _, ptr = rand_batch(sizes, num_features=ch // 2, seed=ch + k)
    x = x * 2 - 1
    lfa = LocalFeatureAggregation(ch)
    fill_params_deterministic(lfa, ch)
    idx, _ = knn_kdtree(pos, ptr.tolist(), pos, ptr.tolist(), k)
    ei = dense_to_edge_index(idx)
    ref_m = lfa.double().train()
    xr = x.double().requires_grad_(True)
    ref = ref_m.aggregate(ei, xr, pos.double())
    gy = torch.from_numpy(np.random.RandomState(ch).uniform(-1, 1, tuple(ref.shape)))
    ref.backward(gy)
    g = LocalFeatureAggregation(ch)
    fill_params_deterministic(g, ch)
    g = g.to(device).train()
    enc_lin, enc_bn = g.mlp_encoder.lins[0], g.mlp_encoder.norms[0].module
    w_att = g.mlp_attention.lins[0].weight
    pos4 = ops.pad_pos(pos.to(device))
    idx32 = idx.to(torch.int32).to(device)
    num_edges = sum(m * min(k, m) for m in sizes)
    mom = ops.lfa_moments(pos4, idx32)
    xg = x.to(device).requires_grad_(True)
    out = ops.LFATrainFn.apply(xg, pos4, idx32, mom, num_edges, enc_lin.weight, enc_lin.bias, enc_bn.weight, enc_bn.bias,
                               enc_lin, enc_bn, w_att, None, True)
    out.backward(gy.float().to(device))
    _close(f"lfa_bf16.out(ch={ch})", out, ref, 0.0, 2e-2)
    _relclose("lfa_bf16.out", out, ref, 1e-2)
    _relclose("lfa_bf16.dx", xg.grad, xr.grad, 3e-2)
    _relclose("lfa_bf16.dW_att", w_att.grad, ref_m.mlp_attention.lins[0].weight.grad, 3e-2)
    _relclose("lfa_bf16.dW_enc", enc_lin.weight.grad, ref_m.mlp_encoder.lins[0].weight.grad, 3e-2)
    _relclose("lfa_bf16.dgamma_enc", enc_bn.weight.grad, ref_m.mlp_encoder.norms[0].module.weight.grad, 3e-2)
    # and it IS a different arithmetic from the fp32 kernels (the bf16 path really ran)
    xg2 = x.to(device).requires_grad_(True)
    out32 = ops.LFATrainFn.apply(xg2, pos4, idx32, mom, num_edges, enc_lin.weight, enc_lin.bias, enc_bn.weight,
                                 enc_bn.bias, enc_lin, enc_bn, w_att, None, False)
    assert not torch.equal(out32, out)
    assert (out32 - out).abs().max().item() < 5e-2


# ----------------------------------------------------------------------------------------------- interpolation
@pytest.mark.parametrize("k", [10, 3, 1])
def test_knn_interpolate_dropin(device, k):
    import myria3d_amd
    from oracle.randla_oracle import knn_interpolate as ref_interp

    rs = np.random.RandomState(k)
    nx, ny = [400, 250], [1500, 900]
    pos_x = torch.from_numpy(rs.uniform(0, 1, (sum(nx), 3)).astype(np.float32))
    pos_y = torch.from_numpy(rs.uniform(0, 1, (sum(ny), 3)).astype(np.float32))
    pos_y[:50] = pos_x[:50]  # coincident points: w = 1/1e-16 path
    x = torch.from_numpy(rs.uniform(-3, 3, (sum(nx), 7)).astype(np.float32))
    bx = torch.repeat_interleave(torch.arange(2), torch.tensor(nx))
    by = torch.repeat_interleave(torch.arange(2), torch.tensor(ny))
    ref = ref_interp(x.double(), pos_x, pos_y, [0, 400, 650], [0, 1500, 2400], k)
    got = myria3d_amd.knn_interpolate(x.to(device), pos_x.to(device), pos_y.to(device), bx.to(device), by.to(device), k=k)
    _close(f"knn_interpolate(k={k})", got, ref, 1e-4, 1e-5)
    # scatter_sum drop-in (myria3d/models/interpolation.py:116)
    index = torch.from_numpy(rs.randint(0, 300, (sum(ny),)))
    out = myria3d_amd.scatter_sum(got, index.to(device), dim=0, out=torch.zeros(300, 7, device=device))
    ref_s = torch.zeros(300, 7, dtype=torch.float64).index_add_(0, index, got.cpu().double())
    _close("scatter_sum", out, ref_s, 1e-5, 1e-4)


@pytest.mark.parametrize("C", [6, 7, 12, 40])
def test_device_interpolator_matches_reference_arithmetic(device, C):
    """Interpolator.reduce_predicted_logits / reduce_predictions_and_save (interpolation.py:98-164) on the device:
    overlapping tiles (points predicted twice), points never predicted, exact ties for argmax, saturated rows."""
    import myria3d_amd
    from oracle.randla_oracle import interpolator_reduce

    rs = np.random.RandomState(C)
    nb_points = 5000
    sizes = [1200, 900, 1500]
    logits_list = [torch.from_numpy(rs.normal(0, 3, (m, C)).astype(np.float32)) for m in sizes]
    logits_list[0][:40] = 0.0                      # all classes tie: argmax must take the first
    logits_list[1][:40, 1] = 60.0                  # saturated softmax: p = 1 -> entropy clamp path
    logits_list[2][:40, 2] = logits_list[2][:40, 4] = 9.5   # two-way tie
    idx_list = [rs.choice(nb_points, m, replace=False) for m in sizes]   # overlaps across tiles, none inside one
    rows, probas, preds, entropy, idx = interpolator_reduce(logits_list, idx_list, nb_points)

    itp = myria3d_amd.DeviceInterpolator()
    for l, i in zip(logits_list[:2], idx_list[:2]):
        itp.store_predictions(l.to(device), [i])               # list of numpy arrays, as the reference collates
    itp.store_predictions(logits_list[2].to(device), torch.from_numpy(idx_list[2]))
    got_rows, got_idx = itp.reduce_predicted_logits(nb_points)
    assert torch.equal(got_idx.cpu(), idx)
    _close("reduced logits", got_rows, rows.double(), 1e-6, 1e-6)

    for l, i in zip(logits_list, idx_list):
        itp.store_predictions(l.to(device), [i])
    out = itp.reduce_predictions(nb_points)
    _close("probas", out["probas"], probas.double(), 1e-6, 1e-5)
    _close("entropy", out["entropy"], entropy.double(), 2e-6, 1e-5)
    # argmax: identical wherever the reference's maximum is unique at fp32 resolution of the merged logits
    top2 = rows.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(out["preds"].cpu()[clear], preds[clear])
    once = torch.bincount(idx, minlength=nb_points)[idx] == 1          # single prediction: merged logits are bit-equal
    ties = (top2[:, 0] == top2[:, 1]) & once
    assert ties.sum() >= 40
    assert torch.equal(out["preds"].cpu()[ties], preds[ties]), "exact ties: first maximum, like torch.argmax"
    assert torch.equal(out["idx_in_full_cloud"].cpu(), idx)
    # function-level: all rows, no index; and the class-code mapping of the reference (interpolation.py:52-56)
    p2, c2, e2 = myria3d_amd.predict_reduce(got_rows)
    _close("probas (no index)", p2, probas.double(), 1e-6, 1e-5)
    mapper = {c: 10 * c + 1 for c in range(C)}
    itp2 = myria3d_amd.DeviceInterpolator(reverse_mapper=mapper)
    itp2.store_predictions(logits_list[0].to(device), [idx_list[0]])
    out2 = itp2.reduce_predictions(nb_points)
    assert torch.equal(out2["preds"].cpu(), torch.argmax(logits_list[0], dim=1) * 10 + 1)


@pytest.mark.parametrize("mode", ["fp32", "bf16_storage", "bf16_operands"])
def test_batched_weight_gradients_every_tile_class_seam_and_row_map(device, mode):
    """Round 6 rewrote the row loops of the weight-gradient kernels (``wgrad_trips_vec`` / ``wgrad_trips_bf16``: row numbers one
    trip ahead, every load of a trip before the first use, ONE X load per step where a tile lies wholly in x0 or x1, the
    64 x 64-tile waves meeting in LDS, XCD-aware workgroup order of the batched launch).  The network's shapes exercise only
    some of the paths (every concatenated input of its 16-tile jobs has its seam on a tile border): this batch holds all seven
    tile classes, seams on and OFF tile borders with and without a row map on x0, ragged row counts (not a multiple of 4 / 32),
    a job with a single row, in one ``m3d_linear_wgrad_batch`` call per activation layout, against fp64."""
    from myria3d_amd import ops

    rs = np.random.RandomState(7)
    dt = torch.bfloat16 if mode == "bf16_storage" else torch.float32
    # (M, N, k0, k1, rows?)                 tile class (TN, TK) / what it adds
    cases = [(3203, 128, 192, 0, False),   # (4, 4) one operand
             (3203, 128, 128, 64, True),   # (4, 4) seam on a tile border, x0 through a row map
             (1601, 64, 96, 96, True),     # (4, 4) seam INSIDE a tile (two X loads per step), row map
             (1601, 64, 96, 96, False),    # ... without
             (5003, 32, 32, 32, True),     # (2, 4) seam inside the tile (the FP1 layer's shape)
             (5003, 64, 32, 0, False),     # (4, 2)
             (4099, 32, 32, 0, False),     # (2, 2)
             (4099, 16, 32, 0, False),     # (1, 2)
             (4099, 32, 9, 0, True),       # (2, 1) fc0's shape: odd K, row map
             (4099, 6, 16, 0, False),      # (1, 1)
             (1, 64, 64, 0, False),        # a single row
             (37, 256, 256, 256, True),    # fewer rows than one 32-row bf16 step has lanes for
             (3203, 128, 128, 0, False),   # whole 128 x 128 blocks (fp32: the LDS-shared form, wgrad_lds_body), ragged last trip
             (1601, 256, 256, 128, True),  # ... seam on a block border, x0 through a row map
             (12800, 256, 128, 0, False),  # ... many row slices
             (31, 512, 512, 0, False)]     # ... less than one 32-row trip
    side = ops.GradSideStream(device)
    jobs, refs = [], []
    for M, N, k0, k1, mapped in cases:
        n_src = M // 3 + 1 if mapped else M
        dz = torch.from_numpy(rs.uniform(-1, 1, (M, N)).astype(np.float32)).to(device).to(dt)
        x0 = torch.from_numpy(rs.uniform(-1, 1, (n_src, k0)).astype(np.float32)).to(device).to(dt)
        x1 = torch.from_numpy(rs.uniform(-1, 1, (M, k1)).astype(np.float32)).to(device).to(dt) if k1 else None
        rows = torch.from_numpy(rs.randint(0, n_src, (M,)).astype(np.int32)).to(device) if mapped else None
        sink = torch.full((N, k0 + k1), 0.5, dtype=torch.float32, device=device)
        a = x0.double()[rows.long()] if mapped else x0.double()
        if k1:
            a = torch.cat([a, x1.double()], 1)
        refs.append(0.5 + dz.double().t() @ a)
        jobs.append((dz, x0, k0, rows, x1, k1, sink, mode == "bf16_operands"))
        side.defer(jobs[-1])
    side.join()
    torch.cuda.synchronize()
    for (M, N, k0, k1, mapped), job, ref in zip(cases, jobs, refs):
        name = f"wgrad batch {mode} M={M} N={N} k0={k0} k1={k1} rows={mapped}"
        if mode == "bf16_operands":
            # products of bf16-rounded operands, fp32 accumulate: relative to the column scale sqrt(M)
            _close(name, job[6], ref, 2e-2, 2e-2 * np.sqrt(M))
        else:
            _close(name, job[6], ref, 1e-5, 3e-6 * np.sqrt(M) + 2e-7 * M)
