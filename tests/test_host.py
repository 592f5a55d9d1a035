"""CPU: host-side logic of the drop-in boundary (no kernel launches)."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from tests.conftest import ROOT


def test_state_dict_is_checkpoint_compatible_with_the_reference_layout():
    from myria3d_amd import HipRandLANet
    from oracle.randla_oracle import RandLANetOracle

    a = HipRandLANet(9, 7).state_dict()
    b = RandLANetOracle(9, 7).state_dict()  # reproduces PyG's key names (tests/test_oracle.py pins them)
    assert list(a) == list(b)
    assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    net = HipRandLANet(9, 7)
    net.load_state_dict({k: v.clone() for k, v in b.items()})  # strict


def test_constructor_and_forward_surface():
    import inspect

    from myria3d_amd import HipRandLANet

    sig = inspect.signature(HipRandLANet.__init__)
    assert list(sig.parameters)[1:] == ["num_features", "num_classes", "decimation", "num_neighbors", "return_logits"]
    assert sig.parameters["decimation"].default == 4 and sig.parameters["num_neighbors"].default == 16
    assert sig.parameters["return_logits"].default is False
    fwd = list(inspect.signature(HipRandLANet.forward).parameters)
    assert fwd[1:5] == ["x", "pos", "batch", "ptr"]
    net = HipRandLANet(9, 6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(10, 9), torch.rand(10, 3), torch.zeros(10, dtype=torch.long), torch.tensor([0, 10]))
    bad = HipRandLANet(9, 6, decimation=0.5)
    with pytest.raises(ValueError, match="higher than"):
        bad(torch.rand(10, 9), torch.rand(10, 3), None, torch.tensor([0, 10]))
    # the function-level drop-ins of the predict path refuse host tensors too
    import myria3d_amd
    for fn, args in [(myria3d_amd.knn_interpolate, (torch.rand(4, 2), torch.rand(4, 3), torch.rand(6, 3))),
                     (myria3d_amd.scatter_sum, (torch.rand(4, 2), torch.zeros(4, dtype=torch.long))),
                     (myria3d_amd.predict_reduce, (torch.rand(4, 6),))]:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            fn(*args)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        myria3d_amd.DeviceInterpolator().store_predictions(torch.rand(4, 6), [np.arange(4)])
    T = myria3d_amd.transforms
    ptr = torch.tensor([0, 4])
    for fn, args in [(T.grid_sampling, (torch.rand(4, 3), None, None, ptr, 0.25)),
                     (T.node_budget, (torch.rand(4, 3), None, None, ptr, 300, 40000)),
                     (T.normalize_tiles, (torch.rand(4, 3), None, ptr))]:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            fn(*args)
    # same Hydra-facing names and constructor arguments as the reference's transform targets
    assert repr(T.GridSampling(0.25)) == "GridSampling(size=0.25)" and T.MinimumNumNodes(300).num == 300
    assert T.MaximumNumNodes(40000).num == 40000 and T.NormalizePos(subtile_width=50).kw["subtile_width"] == 50.0


def test_level_plan_matches_reference_decimation_rule():
    from myria3d_amd import make_plan

    plan = make_plan([0, 12800, 12850, 12851], 4, 16, "cpu")
    assert plan.sizes == [[12800, 50, 1], [3200, 12, 1], [800, 3, 1], [200, 1, 1], [50, 1, 1]]
    assert [p.tolist() for p in plan.ptrs][1] == [0, 3200, 3212, 3213]
    assert plan.totals == [12851, 3213, 804, 202, 52]
    assert plan.num_edges[0] == 12800 * 16 + 50 * 16 + 1 and plan.num_edges[2] == 800 * 16 + 9 + 1
    # all levels' ptr vectors are slices of ONE tensor (one upload); on the CPU there is no staging buffer and no event
    assert [p.tolist() for p in plan.ptrs] == [[0, 12800, 12850, 12851], [0, 3200, 3212, 3213], [0, 800, 803, 804],
                                               [0, 200, 201, 202], [0, 50, 51, 52]]
    base = plan.ptrs[0].data_ptr()
    assert [p.data_ptr() - base for p in plan.ptrs] == [32 * l for l in range(5)] and all(p.is_contiguous() for p in plan.ptrs)
    assert plan.staging is None and plan.ready is None
    from myria3d_amd.randla import plan_ready

    plan_ready(plan)  # nothing to wait for: returns without touching a device
    assert make_plan([0], 4, 16, "cpu").totals == [0, 0, 0, 0, 0]  # an empty batch still has its five (empty) levels


def test_attention_weight_packing_is_the_mfma_b_fragment_order():
    from myria3d_amd.ops import pack_attention_weight

    for ch in (8, 16, 64):
        w = torch.arange(ch * ch, dtype=torch.float32).view(ch, ch)
        p = pack_attention_weight(w)
        chp = max(ch, 16)
        s4n = chp // 16
        flat = p.reshape(-1)
        wp = torch.zeros(chp, chp)
        wp[:ch, :ch] = w
        for nt in range(chp // 16):
            for s4 in range(s4n):
                for lane in (0, 5, 17, 63):
                    for i in range(4):
                        got = flat[((nt * s4n + s4) * 64 + lane) * 4 + i]
                        assert got == wp[16 * nt + (lane & 15), 4 * (4 * s4 + i) + (lane >> 4)]


def test_registration_shim_uses_the_reference_class_factory(monkeypatch):
    """myria3d.models.model.get_neural_net_class picks the first MODEL_ZOO class whose __name__ contains the
    requested string (myria3d/models/model.py:15-29); emulate that module since myria3d's deps are absent."""
    from myria3d_amd import HipRandLANet, register_in_model_zoo

    class PyGRandLANet:  # stand-in
        pass

    fake = types.ModuleType("myria3d.models.model")
    fake.MODEL_ZOO = [PyGRandLANet]

    def get_neural_net_class(class_name):
        for c in fake.MODEL_ZOO:
            if class_name in c.__name__:
                return c
        raise KeyError(f"Unknown class name {class_name}")

    fake.get_neural_net_class = get_neural_net_class
    pkg, models = types.ModuleType("myria3d"), types.ModuleType("myria3d.models")
    pkg.models, models.model = models, fake
    monkeypatch.setitem(sys.modules, "myria3d", pkg)
    monkeypatch.setitem(sys.modules, "myria3d.models", models)
    monkeypatch.setitem(sys.modules, "myria3d.models.model", fake)
    assert register_in_model_zoo() and register_in_model_zoo()
    assert fake.MODEL_ZOO.count(HipRandLANet) == 1
    assert fake.get_neural_net_class("HipRandLANet") is HipRandLANet
    assert fake.get_neural_net_class("PyGRandLANet") is PyGRandLANet
    from myria3d_amd import HipPointNet2

    # (substring matching, model.py:26-29: "PointNet" now finds the set-abstraction variant — BASELINE configs[4])
    assert fake.get_neural_net_class("HipPointNet2") is HipPointNet2 and fake.get_neural_net_class("PointNet") is HipPointNet2
    with pytest.raises(KeyError):
        fake.get_neural_net_class("KPConv")


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "myria3d_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_tile_sharding_is_a_partition():
    from myria3d_amd.ddp import shard_tiles

    for tiles, world in ((128, 8), (16, 1), (10, 4), (3, 8)):
        got = [t for r in range(world) for t in shard_tiles(tiles, r, world)]
        assert got == list(range(tiles))


def test_bench_gpus_flag_starts_that_many_ranks():
    """``python bench.py --gpus 2`` (no launcher, no WORLD_SIZE) must come up as TWO processes in one process group;
    --dry-run-gloo does the launch + rendezvous + one all-reduce without touching a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-gloo"],
                         capture_output=True, text=True, env=env, timeout=240)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    rec = __import__("json").loads(line)
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and len(set(rec["pids"])) == 2


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run-gloo"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE=1" in (res.stdout + res.stderr)


def _free_port() -> int:
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


_FLAT_DDP_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from myria3d_amd import FusedAdam, HipRandLANet
from myria3d_amd.ddp import broadcast_module_state, shard_tiles
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(rank)                                   # different initial weights per rank
net = HipRandLANet(9, 6, return_logits=True).flatten_parameters()
keys = list(net.state_dict().keys())
assert net.flat_parameters.numel() >= 1113686 and "block1.lfa1.mlp_attention.lins.0.weight" in keys
assert all(p.data_ptr() >= net.flat_parameters.data_ptr() for p in net.parameters())     # views of ONE bucket
broadcast_module_state(net)                               # one broadcast of the flat bucket (+ the BN buffers)
got = [torch.zeros_like(net.flat_parameters) for _ in range(world)]
dist.all_gather(got, net.flat_parameters)
assert torch.equal(got[0], got[1]) and torch.equal(net.fc0.weight.reshape(-1), got[0][:net.fc0.weight.numel()])
# what FusedAdam.step(all_reduce=True) does before its single launch: ONE all-reduce(SUM) of the flat gradient,
# scaled by 1/world inside the update
for i, p in enumerate(net.parameters()):
    p.grad.fill_(float(rank + 1) * (i + 1))               # p.grad is a view of net.flat_grads
local = net.flat_grads.clone()
opt = FusedAdam(net, lr=1e-3, all_reduce=True)            # the product path's collective: FusedAdam.reduce_gradients()
scale = opt.reduce_gradients()                            # = what step() does before its single update launch
assert scale == 1.0 / world and opt.uses_collective()
assert not FusedAdam(net, lr=1e-3, all_reduce=False).uses_collective()   # (no exchange asked for: none made)
for i, p in enumerate(net.parameters()):
    assert torch.all(p.grad == 3.0 * (i + 1)), i          # ranks 1 + 2
assert torch.equal(net.flat_grads, local * 3.0 / (rank + 1))
assert list(shard_tiles(32, rank, world)) == list(range(16 * rank, 16 * rank + 16))
dist.destroy_process_group()
print("ok", rank)
"""


def test_flat_bucket_broadcast_and_allreduce_world_size_2_gloo(tmp_path):
    """The N > 1 host logic on the real module: flat parameter / gradient buckets, one broadcast, one all-reduce."""
    script = tmp_path / "flat_ddp_check.py"
    script.write_text(_FLAT_DDP_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), str(script), ROOT],
        capture_output=True, text=True, env=env, timeout=240)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("ok") == 2


def test_forced_collective_on_a_one_rank_group_gloo():
    """``FusedAdam(force_collective=True)``: the all-reduce runs even on a 1-rank group (how the 1-GPU box exercises the
    N > 1 code path, ``bench.py --force-collective``); without it a 1-rank group makes no collective."""
    import torch
    import torch.distributed as dist

    from myria3d_amd import FusedAdam, HipRandLANet

    assert not dist.is_initialized()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        net = HipRandLANet(9, 6, return_logits=True).flatten_parameters()
        assert not FusedAdam(net, lr=1e-3, all_reduce=True).uses_collective()
        opt = FusedAdam(net, lr=1e-3, all_reduce=True, force_collective=True)
        assert opt.uses_collective()
        net.flat_grads.fill_(2.0)
        assert opt.reduce_gradients() == 1.0 and float(net.flat_grads.min()) == 2.0
    finally:
        dist.destroy_process_group()


def test_knn_f64_key_trick_preserves_the_total_order():
    """knn.hip keeps the top-k keys (fp32 d2 bits << 32 | row) as IEEE doubles so that a sorted insertion is a
    v_min_f64 / v_max_f64 chain.  The high word is biased by 0x00100000: every fp32 pattern up to 0x7FDFFFFF (zero,
    denormals, normals, inf, the canonical NaN 0x7FC00000 the hardware produces) must map to a finite NORMAL double, and the doubles must order exactly like the
    64-bit integers; +inf is the empty-slot sentinel above all of them."""
    rs = np.random.RandomState(0)
    d2 = np.concatenate([
        np.array([0.0, 1e-45, 1e-39, 1.17549435e-38, 1.0, 3.4028235e38, np.inf], np.float32),
        np.abs(rs.standard_cauchy(5000)).astype(np.float32), rs.uniform(0, 1e-30, 2000).astype(np.float32),
        np.repeat(np.float32(0.25), 64)])                       # ties in distance: the row decides
    bits = d2.view(np.uint32).astype(np.uint64)
    bits = np.concatenate([bits, np.array([0x7FC00000, 0x7FDFFFFF], np.uint64)])   # canonical NaN; largest pattern covered
    rows = rs.randint(0, 2 ** 31 - 1, bits.shape[0]).astype(np.uint64)
    ukey = (bits << np.uint64(32)) | rows
    dkey = (((bits + np.uint64(0x00100000)) << np.uint64(32)) | rows).view(np.float64)
    assert np.isfinite(dkey).all()
    exp = (dkey.view(np.uint64) >> np.uint64(52)) & np.uint64(0x7FF)
    assert (exp > 0).all() and (exp < 0x7FF).all(), "normal doubles only: min/max return their operands unchanged"
    order_u, order_d = np.argsort(ukey, kind="stable"), np.argsort(dkey, kind="stable")
    assert np.array_equal(order_u, order_d)
    assert len(np.unique(ukey)) == len(np.unique(dkey))
    empty = np.array([0x7FF0000000000000], np.uint64).view(np.float64)[0]
    assert np.isinf(empty) and (dkey < empty).all()
    # decoding gives back the distance bits and the row
    back = dkey.view(np.uint64)
    assert np.array_equal((back >> np.uint64(32)) - np.uint64(0x00100000), bits)
    assert np.array_equal(back & np.uint64(0xFFFFFFFF), rows)


def test_fused_adam_state_dict_speaks_torch_adam(tmp_path):
    """Adam moments and the step counter must survive ``state_dict()`` -> ``load_state_dict()`` (Lightning
    checkpoints / ``ckpt_path`` resume), in ``torch.optim.Adam``'s own layout so that either optimizer can resume the
    other's run.  Host-side logic only (the update kernel itself is covered by tests/test_gpu_train.py)."""
    import torch
    from myria3d_amd import FusedAdam, HipRandLANet

    torch.manual_seed(0)
    net = HipRandLANet(9, 6, return_logits=True).flatten_parameters()
    opt = FusedAdam(net, lr=2e-3, betas=(0.8, 0.95))
    opt.exp_avg.copy_(torch.randn_like(opt.exp_avg))
    opt.exp_avg_sq.copy_(torch.rand_like(opt.exp_avg_sq))
    opt.step_count.fill_(7.0)
    sd = opt.state_dict()
    n_params = len(list(net.parameters()))
    assert sorted(sd["state"].keys()) == list(range(n_params)) and sd["param_groups"][0]["params"] == list(range(n_params))
    assert float(sd["state"][3]["step"]) == 7.0 and sd["param_groups"][0]["betas"] == (0.8, 0.95)
    torch.save(sd, tmp_path / "opt.pt")
    # (a) torch.optim.Adam accepts it verbatim
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in net.parameters()], lr=1.0)
    ref.load_state_dict(torch.load(tmp_path / "opt.pt"))
    for i, (p, q) in enumerate(zip(ref.param_groups[0]["params"], net.parameters())):
        assert torch.equal(ref.state[p]["exp_avg"], sd["state"][i]["exp_avg"]) and float(ref.state[p]["step"]) == 7.0
    assert ref.param_groups[0]["lr"] == 2e-3
    # (b) a fresh FusedAdam restores moments + step from torch.optim.Adam's own state_dict
    net2 = HipRandLANet(9, 6, return_logits=True).flatten_parameters()
    opt2 = FusedAdam(net2, lr=1.0)
    opt2.load_state_dict(ref.state_dict())
    assert float(opt2.step_count) == 7.0 and opt2.param_groups[0]["lr"] == 2e-3
    for (p, off, n) in opt._slices():
        assert torch.equal(opt2.exp_avg[off:off + n], opt.exp_avg[off:off + n])
        assert torch.equal(opt2.exp_avg_sq[off:off + n], opt.exp_avg_sq[off:off + n])
    # one group, no frozen parameters
    with pytest.raises(ValueError):
        opt.add_param_group({"params": [torch.nn.Parameter(torch.zeros(1))]})
    net3 = HipRandLANet(9, 6)
    net3.fc0.weight.requires_grad_(False)
    with pytest.raises(ValueError):
        FusedAdam(net3)


def test_eval_cache_notices_parameter_updates_through_a_parent_module():
    """folded-BatchNorm / packed-weight caches of the eval path must not survive a parent module's
    ``load_state_dict`` (Lightning: ``Model.model``), an optimizer step or an in-place copy."""
    import torch
    from myria3d_amd import HipRandLANet

    class Shell(torch.nn.Module):          # stands in for myria3d.models.model.Model
        def __init__(self):
            super().__init__()
            self.model = HipRandLANet(9, 6)

    shell = Shell().eval()
    net = shell.model
    bn = net.block1.mlp1.norms[0].module
    calls = []
    fold = lambda: calls.append(1) or ("folded", len(calls))
    deps = net._bn_deps(bn)
    assert net._cached(("bn", id(bn)), fold, deps) == ("folded", 1)
    assert net._cached(("bn", id(bn)), fold, deps) == ("folded", 1)          # hit
    shell.load_state_dict(shell.state_dict())                               # parent-level load -> post hook
    assert net._cached(("bn", id(bn)), fold, deps) == ("folded", 2)
    with torch.no_grad():
        bn.running_var.mul_(2.0)                                            # in-place update: version counter moves
    assert net._cached(("bn", id(bn)), fold, deps) == ("folded", 3)
    torch.optim.SGD([bn.weight], lr=0.1)                                    # an optimizer step in eval mode
    with torch.no_grad():
        bn.weight.add_(1.0)
    assert net._cached(("bn", id(bn)), fold, deps) == ("folded", 4)
    assert net._cached(("bn", id(bn)), fold, deps) == ("folded", 4)


def test_zero_arena_learns_its_size_across_an_eval_pass():
    """ops.ZeroArena: the training step's accumulation targets come from ONE zero fill once the size is known — also when
    an eval pass (arena.stop()) runs between two training steps, as in bench.py and in any train/validate loop."""
    from myria3d_amd.ops import ZeroArena

    dev = torch.device("cpu")
    ar = ZeroArena()
    ar.begin(dev)
    a = ar.zeros((3, 5), torch.float32, dev)
    b = ar.zeros((7,), torch.float64, dev)
    assert ar.buf is None and a.abs().sum() == 0 and b.abs().sum() == 0   # first step: individual fills
    ar.stop()                                                              # eval pass
    assert ar.need >= 3 * 5 * 4 + 7 * 8
    ar.begin(dev)
    a = ar.zeros((3, 5), torch.float32, dev)
    b = ar.zeros((7,), torch.float64, dev)
    base = ar.buf.data_ptr()
    assert a.data_ptr() == base and b.data_ptr() == base + 256            # views of the one buffer, 256-byte spans
    assert a.dtype == torch.float32 and b.dtype == torch.float64 and b.shape == (7,)
    c = ar.zeros((1000,), torch.float32, dev)                              # more than last time: falls back, and is learned
    assert c.data_ptr() < base or c.data_ptr() >= base + ar.buf.numel()
    ar.begin(dev)
    assert ar.buf.numel() >= 512 + 4096


def test_node_budget_offsets_on_the_host_follow_the_device_rule():
    """``transforms.node_budget_offsets`` (round 5: the predict chain keeps the CSR offsets of a batch on the host so that the net's
    level plan needs no device read-back; round 6: ONE function, which ``node_budget`` itself sizes its outputs with) = the reference's
    ``MinimumNumNodes`` + ``MaximumNumNodes`` (transforms.py:48-87): tiles with 0 points stay empty, tiles below the minimum
    are filled up to it, tiles above the maximum are cut."""
    import torch

    from myria3d_amd.transforms import node_budget_offsets as _budget_offsets

    counts = [0, 1, 299, 300, 301, 39999, 40000, 40001, 123456]
    ptr = [0]
    for c in counts:
        ptr.append(ptr[-1] + c)
    for minimum, maximum in ((300, 40000), (0, 40000), (300, None), (0, None)):
        t = torch.tensor(counts)
        out = t.clone()
        if minimum:
            out = torch.where((t > 0) & (t < minimum), torch.full_like(out, minimum), out)
        if maximum is not None:
            out = out.clamp(max=maximum)
        want = [0] + out.cumsum(0).tolist()
        assert _budget_offsets(ptr, minimum, maximum) == want, (minimum, maximum)


def test_pending_batchnorm_bookkeeping():
    """``ops.PendingBN`` (BatchNorm apply-on-load, round 5): the producer hands over an allocated, unwritten activation buffer;
    the consumer recognises it by identity or by storage, and a forward that ends with one left over would settle it (here only
    the bookkeeping, on host tensors: the launches need the GPU)."""
    import torch

    from myria3d_amd import ops

    assert ops.take_pending(None) is None and ops.take_pending(torch.zeros(3)) is None
    z, y = torch.zeros(5, 4), torch.empty(5, 4)
    p = ops.PendingBN(z, torch.zeros(1, 2, 4, dtype=torch.float64), 5, torch.nn.BatchNorm1d(4), True, y,
                      tuple(torch.empty(4) for _ in range(4)))
    ops._pending.append(p)
    try:
        assert ops.take_pending(y) is p
        assert ops.take_pending(y.view(5, 4)) is p          # another tensor object over the same storage and shape
        assert ops.take_pending(torch.empty(5, 4)) is None  # somebody else's buffer
        assert ops.take_pending(z) is None                  # the raw output is not the activation
        p.done = True                                       # (as a fused consumer leaves it)
        assert p.materialize() is y                         # no launch once it is done
    finally:
        ops._pending.clear()


def test_edge_row_backward_is_offered_for_the_narrow_complete_layers_only():
    """Host-side decisions of round 5's atomic-free input gradient (no launch): ``m3d_lfa_bwd_edge_rows_ok`` says where
    ``m3d_lfa_bwd(flags | 32)`` stores per-edge rows (8 / 16 channels, K = 16, 32-bit byte offsets, LeakyReLU slope in [0, 1]),
    and the reverse-list builder sizes its workspace (ranks [n K], counts [n], block sums)."""
    from myria3d_amd import _lib

    lib = _lib.lib()
    ok = lib.m3d_lfa_bwd_edge_rows_ok
    assert ok(204800, 16, 8, 0.2) == 1 and ok(204800, 16, 16, 0.2) == 1
    assert ok(204800, 16, 32, 0.2) == 0 and ok(204800, 16, 64, 0.2) == 0
    assert ok(204800, 32, 16, 0.2) == 0 and ok(204800, 8, 16, 0.2) == 0
    assert ok(0, 16, 16, 0.2) == 0 and ok(204800, 16, 16, -0.1) == 0 and ok(204800, 16, 16, 1.01) == 0
    assert ok((1 << 31) // (16 * 8 * 4), 16, 16, 0.2) == 0 and ok((1 << 31) // (16 * 8 * 4) - 1, 16, 16, 0.2) == 1
    wsb = lib.m3d_knn_reverse_workspace_bytes
    assert wsb(204800, 16) >= 204800 * 16 * 4 + 204800 * 4 + (204800 // 4096 + 1) * 4
    assert wsb(-1, 16) == 0 and wsb(10, 0) == 0
