"""GPU parity of the whole hot path: ``HipRandLANet`` (HIP kernels through the C ABI) vs the CPU oracle.

Tolerances (fp32 both sides, different summation orders through ~45 GEMMs / 35 BatchNorms):
  eval logits      |d| <= 1e-4 + 1e-4*|ref|      and identical argmax on >= 99.99 % of the points whose top-2 logits are not
                   tied inside that tolerance (SURVEY 8c's numbers, asserted since round 5; recorded intermediates 2e-4)
  train logits     |d| <= 1e-3 + 1e-3*|ref|   (the eval bound x 10; asserted since round 6, measured <= 1.5e-4)
  parameter grads  relative L2 error vs an fp64 oracle run <= max(1e-3, 2 x the error of the fp32 ORACLE against the same fp64
                   run) per tensor: 1e-3 is SURVEY 8c's number; where fp32 arithmetic itself cannot hold it (cancelling sums:
                   the reference's own fp32 run is 4.2e-3 off its fp64 run in block2.lfa2.mlp_encoder.norms.0.module.bias) the
                   bound is what the reference's arithmetic achieves, measured in the same test and printed beside ours
"""
import os

import numpy as np
import pytest
import torch

from tests._util import fill_params_deterministic, rand_batch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "randla_small.npz")


def _pair(device, num_features=9, num_classes=6, k=16, seed=0, return_logits=True):
    from myria3d_amd import HipRandLANet
    from oracle.randla_oracle import RandLANetOracle

    ref = RandLANetOracle(num_features, num_classes, num_neighbors=k, return_logits=return_logits)
    fill_params_deterministic(ref, seed)
    net = HipRandLANet(num_features, num_classes, num_neighbors=k, return_logits=return_logits)
    net.load_state_dict(ref.state_dict())
    return ref, net.to(device)


def _report(name, got, ref, rtol, atol):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs()
    used = (err / (atol + rtol * ref.abs())).max().item()  # 1.0 = at the bound
    print(f"[parity] {name}: max_abs_err={err.max().item():.3e} ref_absmax={ref.abs().max().item():.3e} "
          f"fraction of the tolerance used = {used:.3f}")
    bad = err > atol + rtol * ref.abs()
    assert not bool(bad.any()), f"{name}: {int(bad.sum())} / {bad.numel()} outside tol, max err {err.max().item():.3e}"


# SURVEY section 8c's train-mode numbers (round 6: asserted as stated)
TRAIN_RTOL, TRAIN_ATOL, GRAD_TOL = 1e-3, 1e-3, 1e-3


def _grad_table(what, got, ref64, ref32, tol=GRAD_TOL, skip_zero=True):
    """Per parameter tensor: relative L2 error of ``got`` (HIP) and of ``ref32`` (the same arithmetic run in fp32 on the CPU: the
    oracle, or the reference's own module) against ``ref64`` (its fp64 run).  Asserts HIP <= max(tol, 2 x the fp32 run's error);
    prints the worst rows.  Dicts name -> tensor."""
    rows = []
    for name, g64 in ref64.items():
        g64 = g64.detach().cpu().double()
        den = g64.norm().item()
        gh = got[name].detach().cpu().double()
        if (".lins." in name and name.endswith("bias")) or den < 1e-8:
            # a Linear bias in front of a train-mode BatchNorm (and fc0.bias, removed by the BatchNorms of mlp1 / shortcut):
            # analytically zero, both sides hold rounding noise only
            if skip_zero:
                assert gh.abs().max().item() < 1e-5, name
                continue
        e_hip = (gh - g64).norm().item() / max(den, 1e-30)
        e_f32 = (ref32[name].detach().cpu().double() - g64).norm().item() / max(den, 1e-30) if ref32 is not None else float("nan")
        rows.append((e_hip, e_f32, name))
    rows.sort(reverse=True)
    print(f"[parity] {what}: {len(rows)} parameter gradients vs the fp64 run; worst (HIP rel-L2 | fp32 CPU run rel-L2 | tensor):")
    for e_hip, e_f32, name in rows[:6]:
        print(f"[parity]     {e_hip:.3e} | {e_f32:.3e} | {name}")
    over = [(e, f, n) for e, f, n in rows if e > tol]
    print(f"[parity] {what}: {len(over)} tensors above {tol:g}; all of them within 2 x the fp32 CPU run's own error: "
          f"{all(e <= 2 * f for e, f, _ in over)}")
    for e_hip, e_f32, name in rows:
        bound = max(tol, 2 * e_f32) if ref32 is not None else tol
        assert e_hip <= bound, f"grad {name}: HIP {e_hip:.3e} vs fp64, fp32 CPU run {e_f32:.3e}, bound {bound:.3e}"
    return rows


# SURVEY section 8c's own numbers for the eval forward (asserted since round 5 on the FINAL logits of every eval test; the
# recorded intermediates keep 2e-4: they hold raw BatchNorm inputs of magnitude 10 - 100)
EVAL_RTOL, EVAL_ATOL, EVAL_ARGMAX = 1e-4, 1e-4, 0.9999


def _argmax_agreement(name, got, ref, floor=EVAL_ARGMAX):
    """Share of points with the same argmax, counted over the points whose top-2 reference logits are further apart than
    the logit tolerance (a tie inside the tolerance can legitimately fall either way)."""
    got, ref = got.detach().cpu(), ref.detach().cpu().float()
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * (EVAL_ATOL + EVAL_RTOL * top2[:, 0].abs())
    agree_all = (got.argmax(1) == ref.argmax(1)).float().mean().item()
    agree = (got.argmax(1)[clear] == ref.argmax(1)[clear]).float().mean().item() if bool(clear.any()) else 1.0
    print(f"[parity] {name}: argmax agreement {agree_all:.6f} over all points, {agree:.6f} over the {int(clear.sum())} "
          f"points without a tie inside the tolerance")
    assert agree_all >= 0.999 and agree >= floor, (name, agree_all, agree)


@pytest.mark.parametrize("sizes", [[300, 211], [50, 50], [1250, 1000], [5, 1, 40]])
def test_eval_logits_match_oracle(device, sizes):
    from oracle.randla_oracle import fixed_decimation_indices

    ref, net = _pair(device, seed=len(sizes))
    x, pos, batch, ptr = rand_batch(sizes, seed=sum(sizes))
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=1)
    ref.eval(), net.eval()
    rec_r, rec_g = {}, {}
    with torch.no_grad():
        out_r = ref(x, pos, batch, ptr, decimation_idx=dec, record=rec_r)
        out_g = net(x.to(device), pos.to(device), batch.to(device), ptr.to(device), decimation_idx=dec, record=rec_g)
    for key in sorted(rec_g):
        if key.endswith("knn_idx"):
            assert torch.equal(rec_g[key].cpu().long(), rec_r[key]), key
        elif key in rec_r:
            _report(key, rec_g[key], rec_r[key], 2e-4, 2e-4)
    _report("logits", out_g, out_r, EVAL_RTOL, EVAL_ATOL)
    _argmax_agreement("logits", out_g, out_r)


def test_eval_lidar_tiles_and_log_softmax(device):
    from oracle.randla_oracle import fixed_decimation_indices, synthetic_batch

    ref, net = _pair(device, seed=3, return_logits=False)
    x, pos, batch, ptr, _ = synthetic_batch([3000, 2000])
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=2)
    ref.eval(), net.eval()
    with torch.no_grad():
        out_r = ref(x, pos, batch, ptr, decimation_idx=dec)
        out_g = net(x.to(device), pos.to(device), batch.to(device), ptr.to(device), decimation_idx=dec)
    _report("log_probas", out_g, out_r, EVAL_RTOL, EVAL_ATOL)
    assert torch.allclose(out_g.exp().sum(1).cpu(), torch.ones(5000), atol=1e-4)


def test_dense_neighbourhood_k32_matches_oracle(device):
    """BASELINE config 5's RandLA part (K = 32 neighbours): eval logits and a train-mode forward/backward."""
    from oracle.randla_oracle import fixed_decimation_indices

    ref, net = _pair(device, k=32, seed=5)
    sizes = [900, 640]
    x, pos, batch, ptr = rand_batch(sizes, seed=32)
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=6)
    ref.eval(), net.eval()
    with torch.no_grad():
        out_r = ref(x, pos, batch, ptr, decimation_idx=dec)
        out_g = net(x.to(device), pos.to(device), None, ptr.to(device), decimation_idx=dec)
    _report("k32.eval_logits", out_g, out_r, EVAL_RTOL, EVAL_ATOL)
    ref.train(), net.train()
    mask = torch.ones(sum(sizes), 32)
    out_r = ref(x, pos, batch, ptr, decimation_idx=dec, dropout_mask=mask)
    out_g = net(x.to(device), pos.to(device), None, ptr.to(device), decimation_idx=dec, dropout_mask=mask.to(device))
    _report("k32.train_logits", out_g, out_r, TRAIN_RTOL, TRAIN_ATOL)
    out_g.square().mean().backward()
    out_r.square().mean().backward()
    gr = dict(ref.named_parameters())
    for name in ("block1.lfa1.mlp_attention.lins.0.weight", "block3.lfa2.mlp_encoder.lins.0.weight", "fp2.nn.lins.0.weight"):
        a, b = dict(net.named_parameters())[name].grad.cpu().double(), gr[name].grad.double()
        assert (a - b).norm().item() <= 5e-3 * b.norm().item() + 1e-7, name


def test_golden_fixture(device):
    """Committed golden vectors (generated by tests/golden/make_golden.py from the oracle in the build container)."""
    from myria3d_amd import HipRandLANet

    g = np.load(GOLDEN)
    net = HipRandLANet(9, 6, return_logits=True)
    fill_params_deterministic(net, int(g["param_seed"]))
    net = net.to(device).eval()
    dec = [torch.from_numpy(g[f"dec{i}"]) for i in range(4)]
    with torch.no_grad():
        out = net(torch.from_numpy(g["x"]).to(device), torch.from_numpy(g["pos"]).to(device), None,
                  torch.from_numpy(g["ptr"]).to(device), decimation_idx=dec)
    _report("golden.eval_logits", out, torch.from_numpy(g["logits_eval"]), EVAL_RTOL, EVAL_ATOL)
    rec = {}
    net.train()
    out_t = net(torch.from_numpy(g["x"]).to(device), torch.from_numpy(g["pos"]).to(device), None,
                torch.from_numpy(g["ptr"]).to(device), decimation_idx=dec,
                dropout_mask=torch.from_numpy(g["dropout_mask"]).to(device), record=rec)
    assert torch.equal(rec["block1.knn_idx"].cpu().long(), torch.from_numpy(g["knn_idx_level1"]))
    _report("golden.train_logits", out_t, torch.from_numpy(g["logits_train"]), TRAIN_RTOL, TRAIN_ATOL)


def test_reference_fixture_vs_hip_net(device):
    """``tests/golden/randla_reference.npz``: outputs of the reference's OWN ``pyg_randla_net.py`` (run by
    ``tests/golden/make_golden_from_reference.py``; its ``stack`` entry says whether real PyG wheels or
    ``tests/_pyg_stub`` sat under it) against the HIP net directly — no oracle in between: eval logits, train logits,
    loss, parameter gradients, running statistics."""
    from myria3d_amd import HipRandLANet

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "randla_reference.npz"))
    print(f"[parity] reference fixture generated on: {g['stack']}")
    for pre in [str(s) for s in g["sets"]]:  # "" = [700, 333, 50]; "b/" = [300, 9, 1, 120]: K_eff < K and a 1-point cloud
        _reference_fixture_case(device, g, pre)


def _reference_fixture_case(device, g, pre):
    from myria3d_amd import HipRandLANet

    t = lambda k: torch.from_numpy(g[pre + k])
    net = HipRandLANet(9, 6, return_logits=True)
    fill_params_deterministic(net, int(g["param_seed"]))
    net.mlp_classif.dropout = [0.0, 0.0]
    net = net.to(device).eval()
    dec = [t(f"dec{i}") for i in range(4)]
    x, pos, ptr = t("x").to(device), t("pos").to(device), t("ptr").to(device)
    with torch.no_grad():
        out = net(x, pos, None, ptr, decimation_idx=dec)
    _report(f"reference_fixture[{pre}].eval_logits", out, t("logits_eval"), EVAL_RTOL, EVAL_ATOL)
    assert (out.cpu().argmax(1) == t("logits_eval").argmax(1)).float().mean().item() >= 0.999
    net.train()
    out_t = net(x, pos, None, ptr, decimation_idx=dec)
    _report(f"reference_fixture[{pre}].train_logits", out_t, t("logits_train"), TRAIN_RTOL, TRAIN_ATOL)
    loss = torch.nn.functional.cross_entropy(out_t, t("y").to(device))
    assert abs(loss.item() - float(g[pre + "loss_train"])) < 1e-3 * max(1.0, abs(float(g[pre + "loss_train"])))
    loss.backward()
    got = {k: p.grad for k, p in net.named_parameters()}
    own = lambda k: k.startswith(pre) and (pre or not k.startswith("b/"))
    ref32 = {k[len(pre) + 5:]: torch.from_numpy(g[k]) for k in g.files if own(k) and k[len(pre):].startswith("grad:")}
    ref64 = {k[len(pre) + 7:]: torch.from_numpy(g[k]) for k in g.files if own(k) and k[len(pre):].startswith("grad64:")}
    assert len(ref32) == len(ref64) == 152
    # round 6: every gradient against the reference module's own FP64 run; the bound is SURVEY 8c's 1e-3, or twice what the
    # reference's own fp32 run achieves against that yardstick where fp32 cannot hold 1e-3 (rounds 3-5 compared fp32 with fp32
    # at 5e-3 / 2e-2: the 4.25e-3 of round 5's margins log was the REFERENCE's rounding, not the kernels')
    rows = _grad_table(f"reference_fixture[{pre}]", got, ref64, ref32)
    assert len(rows) >= 100
    bufs = dict(net.named_buffers())
    for k in g.files:
        if k.startswith(pre + "buf:") and (pre or not k.startswith("b/")):
            assert torch.allclose(bufs[k[len(pre) + 4:]].cpu(), torch.from_numpy(g[k]), rtol=1e-3, atol=1e-5), k


def _train_parity(device, x, pos, batch, ptr, y, seed, k=16, grad_tol=GRAD_TOL, flat=False, precision="fp32"):
    """One train-mode forward + cross-entropy + backward of HipRandLANet against the fp64 oracle on the same weights,
    decimation indices and dropout mask: logits, loss, every parameter gradient, the running statistics."""
    from oracle.randla_oracle import fixed_decimation_indices

    ref, net = _pair(device, seed=seed, k=k)
    net.matmul_precision = precision
    ref = ref.double()
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=3)
    n = x.shape[0]
    rs = np.random.RandomState(1)
    mask = torch.from_numpy((rs.uniform(size=(n, 32)) > 0.5).astype(np.float32))
    ref.train(), net.train()
    out_r = ref(x.double(), pos.double(), batch, ptr, decimation_idx=dec, dropout_mask=mask.double())
    loss_r = torch.nn.functional.cross_entropy(out_r, y)
    loss_r.backward()
    if flat:
        # the path the bench times: flat parameter / gradient buffers, gradient sinks, weight gradients and LFA reduces
        # deferred to the batched launches at the end of the backward pass (GradSideStream.flush), the HIP criterion
        from myria3d_amd import FusedAdam, cross_entropy

        net.flatten_parameters()
        opt = FusedAdam(net, lr=1e-3)
        assert net.grad_side is not None
    out_g = net(x.to(device), pos.to(device), batch.to(device), ptr.to(device), decimation_idx=dec,
                dropout_mask=mask.to(device))
    if flat:
        loss_g = cross_entropy(out_g, y.to(device), ignore_index=65)
        loss_g.backward()
        opt.reduce_gradients()  # joins the deferred launches (what opt.step() does before the update)
        assert net._use_sinks and all(p.grad.data_ptr() >= net.flat_grads.data_ptr() for p in net.parameters())
    else:
        loss_g = torch.nn.functional.cross_entropy(out_g, y.to(device))
        loss_g.backward()
    _report("train.logits", out_g, out_r, TRAIN_RTOL, TRAIN_ATOL)
    assert abs(loss_g.item() - loss_r.item()) < 1e-3 * max(1.0, abs(loss_r.item()))
    for name, p in net.named_parameters():
        assert p.grad is not None, f"{name} got no gradient (DDP find_unused_parameters=False needs all)"
    # the fp32 column: the ORACLE once more in fp32 on the CPU (same weights, indices, mask) — how far plain fp32 arithmetic
    # lands from the fp64 run, tensor by tensor
    ref32, _ = _pair(device, seed=seed, k=k)
    ref32.train()
    out32 = ref32(x, pos, batch, ptr, decimation_idx=dec, dropout_mask=mask)
    torch.nn.functional.cross_entropy(out32, y).backward()
    rows = _grad_table("train", {n: p.grad for n, p in net.named_parameters()},
                       {n: p.grad for n, p in ref.named_parameters()}, {n: p.grad for n, p in ref32.named_parameters()},
                       tol=grad_tol)
    print(f"[parity] worst parameter-gradient relative L2 error: {rows[0][2]} {rows[0][0]:.3e}")
    # running statistics were updated identically
    got_buffers = dict(net.named_buffers())
    for nr, br in ref.named_buffers():
        if nr.endswith("running_mean") or nr.endswith("running_var"):
            assert torch.allclose(got_buffers[nr].cpu().double(), br, rtol=1e-3, atol=1e-5), nr
    return ref, net, dec


@pytest.mark.parametrize("sizes", [[300, 211], [64, 700]])
def test_train_forward_backward_match_oracle(device, sizes):
    # at these tiny sizes the deepest BatchNorms see 1-4 rows per cloud and amplify fp32 rounding: the bound is
    # max(1e-3, 2 x what the fp32 oracle itself achieves against the fp64 run), tensor by tensor (_grad_table)
    x, pos, batch, ptr = rand_batch(sizes, seed=sizes[0])
    y = torch.from_numpy(np.random.RandomState(1).randint(0, 6, (sum(sizes),)))
    _train_parity(device, x, pos, batch, ptr, y, seed=7)


def test_baseline_tiles_train_and_eval_match_oracle(device):
    """Two full BASELINE-config-2 tiles (12 800 synthetic Lidar-HD-shaped points each, K = 16): the launch shapes the
    bench times — 25 600 / 6 400 / 1 600 / 400 centres per level, every persistent LFA-backward workgroup in its loop
    at level 1 (3 200 groups over 1 024 workgroups) — with NUMBERS against the oracle: train logits 1e-3, every
    parameter gradient max(1e-3, 2 x the fp32 oracle's own error) relative L2 against the fp64 oracle, running statistics;
    then eval logits 1e-4 (fp32 oracle)."""
    from oracle.randla_oracle import synthetic_batch

    x, pos, batch, ptr, y = synthetic_batch([12800, 12800])
    ref, net, dec = _train_parity(device, x, pos, batch, ptr, y, seed=7)
    ref = ref.float().eval()
    net.eval()
    with torch.no_grad():
        out_r = ref(x, pos, batch, ptr, decimation_idx=dec)
        out_g = net(x.to(device), pos.to(device), None, ptr.to(device), decimation_idx=dec)
    _report("baseline_tiles.eval_logits", out_g, out_r, EVAL_RTOL, EVAL_ATOL)
    _argmax_agreement("baseline_tiles.eval_logits", out_g, out_r)


def test_config2_full_batch_eval_logits_match_oracle(device):
    """BASELINE config 2's WHOLE batch — 16 tiles x 12 800 synthetic Lidar-HD-shaped points, K = 16, the launch shapes the
    bench times (204 800 / 51 200 / 12 800 / 3 200 / 800 rows) — eval logits against the CPU oracle on the same decimation
    indices, at SURVEY 8c's tolerance (round 4 had numbers at 2 tiles and properties only at 16)."""
    from oracle.randla_oracle import RandLANetOracle, fixed_decimation_indices, synthetic_batch
    from myria3d_amd import HipRandLANet

    x, pos, batch, ptr, _ = synthetic_batch([12800] * 16)
    ref = RandLANetOracle(9, 6, num_neighbors=16, return_logits=True, knn="kdtree")
    fill_params_deterministic(ref, 16)
    net = HipRandLANet(9, 6, num_neighbors=16, return_logits=True)
    net.load_state_dict(ref.state_dict())
    net = net.to(device).eval()
    ref.eval()
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=12)
    with torch.no_grad():
        out_g = net(x.to(device), pos.to(device), None, ptr.to(device), decimation_idx=dec)
        out_r = ref(x, pos, batch, ptr, decimation_idx=dec)
    assert out_g.shape == (16 * 12800, 6)
    _report("config2_full_batch.eval_logits", out_g, out_r, EVAL_RTOL, EVAL_ATOL)
    _argmax_agreement("config2_full_batch.eval_logits", out_g, out_r)


def test_flattened_path_every_gradient_vs_fp64_oracle(device):
    """The flattened / deferred / batched backward (gradient sinks, ``m3d_linear_wgrad_batch``,
    ``m3d_lfa_bwd_reduce_batch``, slot-mode BatchNorm, HIP cross-entropy) at 2 x 12 800 points: EVERY parameter gradient
    within max(1e-3, 2 x the fp32 oracle's own error) relative L2 of the fp64 oracle."""
    from oracle.randla_oracle import synthetic_batch

    x, pos, batch, ptr, y = synthetic_batch([12800, 12800])
    _train_parity(device, x, pos, batch, ptr, y, seed=7, flat=True)


@pytest.mark.parametrize("num_nodes", [[12500, 12500], [12500, 10000]])
def test_reference_size_cases_numeric(device, num_nodes):
    """The reference's own test sizes and input distribution (tests/myria3d/models/modules/test_randla_nets.py:8-40:
    x, pos ~ U(0,1), F = 9, C = 6, K = 16, decimation 4) — the reference asserts shapes only; here the numbers are
    compared with the oracle (train fwd + bwd, then eval logits)."""
    x, pos, batch, ptr = rand_batch(num_nodes, seed=1)
    y = torch.from_numpy(np.random.RandomState(2).randint(0, 6, (sum(num_nodes),)))
    ref, net, dec = _train_parity(device, x, pos, batch, ptr, y, seed=11)
    ref = ref.float().eval()
    net.eval()
    with torch.no_grad():
        out_r = ref(x, pos, batch, ptr, decimation_idx=dec)
        out_g = net(x.to(device), pos.to(device), None, ptr.to(device), decimation_idx=dec)
    assert out_g.shape == torch.Size([sum(num_nodes), 6])
    _report("reference_sizes.eval_logits", out_g, out_r, EVAL_RTOL, EVAL_ATOL)


def test_dense_tile_40000_points_k32_eval_matches_oracle(device):
    """BASELINE config 5's RandLA part at full tile size (40 000 points, K = 32): eval logits vs the oracle."""
    from oracle.randla_oracle import fixed_decimation_indices, synthetic_batch

    ref, net = _pair(device, k=32, seed=5)
    x, pos, batch, ptr, _ = synthetic_batch([40000])
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=6)
    ref.eval(), net.eval()
    with torch.no_grad():
        out_r = ref(x, pos, batch, ptr, decimation_idx=dec)
        out_g = net(x.to(device), pos.to(device), None, ptr.to(device), decimation_idx=dec)
    _report("dense_tile.eval_logits", out_g, out_r, EVAL_RTOL, EVAL_ATOL)
    assert (out_g.cpu().argmax(1) == out_r.argmax(1)).float().mean().item() >= 0.999


def test_dense_tile_40000_points_k32_train_matches_oracle(device):
    """BASELINE config 5's RandLA part at full tile size, TRAIN mode (40 000 points, K = 32: the 32-neighbour LFA forward /
    backward launch shapes at 40 000 / 10 000 / 2 500 / 625 centres): logits, loss, every parameter gradient and the running
    statistics against the fp64 oracle, plain and with the flat-buffer / deferred-launch path the bench times."""
    from oracle.randla_oracle import synthetic_batch

    x, pos, batch, ptr, y = synthetic_batch([40000])
    _train_parity(device, x, pos, batch, ptr, y, seed=5, k=32, flat=True)


@pytest.mark.parametrize("num_nodes", [[12500, 12500], [50, 50], [12500, 10000]])
def test_reference_shape_cases(device, num_nodes):
    """The reference's own test (tests/myria3d/models/modules/test_randla_nets.py:8-40): default train mode, random
    decimation on device, output shape == (sum N, C); plus finiteness and a backward pass."""
    from myria3d_amd import HipRandLANet

    torch.manual_seed(0)
    net = HipRandLANet(9, 6, decimation=4, num_neighbors=16).to(device)
    x, pos, batch, ptr = rand_batch(num_nodes, seed=1)
    out = net(x.to(device), pos.to(device), batch.to(device), ptr.to(device))
    assert out.shape == torch.Size([sum(num_nodes), 6])
    assert bool(torch.isfinite(out).all())
    out.logsumexp(1).mean().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    # decimation on device: each level keeps max(1, n//4) distinct points per cloud
    for lvl, idx in enumerate(net.last_decimation_idx):
        assert idx.unique().numel() == idx.numel()


def test_full_size_properties(device):
    """BASELINE config 2 shape (B=16 x 12 800, K=16): size-independent properties instead of an O(N^2) oracle."""
    from myria3d_amd import HipRandLANet, ops
    from oracle.randla_oracle import synthetic_batch

    x, pos, batch, ptr, _ = synthetic_batch([12800] * 16)
    posd, ptrd = pos.to(device), ptr.to(device)
    index = ops.KnnIndex(posd, ptrd)
    idx, d2 = index.query(16, qry=index, want_d2=True)
    idx, d2 = idx.cpu().long(), d2.cpu()
    assert bool((d2[:, 0] == 0).all())  # self (or an exact duplicate) comes first
    assert bool((d2[:, 1:] >= d2[:, :-1]).all())  # ascending
    assert bool((idx // 12800 == (torch.arange(idx.shape[0]) // 12800)[:, None]).all())  # never crosses tiles
    # recomputed distances agree bit-for-bit with what the kernel reported
    diff = pos[idx] - pos[:, None, :]
    assert torch.equal((diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2], d2)
    # spot-check 256 random queries against brute force
    from oracle.randla_oracle import knn_exact
    q = torch.randperm(idx.shape[0])[:256]
    for b in range(16):
        sel = q[(q // 12800) == b]
        if sel.numel() == 0:
            continue
        ri, _ = knn_exact(pos[b * 12800:(b + 1) * 12800], [0, 12800], pos[sel], [0, sel.numel()], 16)
        assert torch.equal(ri + b * 12800, idx[sel])
    torch.manual_seed(0)
    net = HipRandLANet(9, 6, return_logits=True).to(device).eval()
    with torch.no_grad():
        a = net(x.to(device), posd, None, ptrd, decimation_idx=None)
    assert a.shape == (16 * 12800, 6) and bool(torch.isfinite(a).all())
    # tiles are independent units: evaluating tile 3 alone gives the same logits (same decimation indices)
    dec = [i.clone() for i in net.last_decimation_idx]
    lvl_sizes = [12800, 3200, 800, 200, 50]
    dec3 = [d[3 * lvl_sizes[l + 1]:4 * lvl_sizes[l + 1]].long() - 3 * lvl_sizes[l] for l, d in enumerate(dec)]
    with torch.no_grad():
        full = net(x.to(device), posd, None, ptrd, decimation_idx=dec)
        one = net(x[3 * 12800:4 * 12800].to(device), posd[3 * 12800:4 * 12800], None,
                  torch.tensor([0, 12800], device=device), decimation_idx=dec3)
    assert torch.allclose(full[3 * 12800:4 * 12800], one, rtol=1e-4, atol=1e-4)


def test_full_size_properties_dense_tiles_k32(device):
    """BASELINE config 5's RandLA part at full size (40 000-point tiles, K = 32): the same size-independent
    properties as config 2, plus one finite training step."""
    import myria3d_amd
    from myria3d_amd import HipRandLANet, ops
    from oracle.randla_oracle import knn_exact, synthetic_batch

    N, B, K = 40000, 4, 32
    x, pos, batch, ptr, y = synthetic_batch([N] * B)
    posd, ptrd = pos.to(device), ptr.to(device)
    index = ops.KnnIndex(posd, ptrd)
    idx, d2 = index.query(K, qry=index, want_d2=True)
    idx, d2 = idx.cpu().long(), d2.cpu()
    assert bool((d2[:, 0] == 0).all()) and bool((d2[:, 1:] >= d2[:, :-1]).all())
    assert bool((idx // N == (torch.arange(idx.shape[0]) // N)[:, None]).all())
    diff = pos[idx] - pos[:, None, :]
    assert torch.equal((diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2], d2)
    q = torch.randperm(idx.shape[0], generator=torch.Generator().manual_seed(5))[:128]
    for b in range(B):
        sel = q[(q // N) == b]
        if sel.numel():
            ri, _ = knn_exact(pos[b * N:(b + 1) * N], [0, N], pos[sel], [0, sel.numel()], K)
            assert torch.equal(ri + b * N, idx[sel])
    torch.manual_seed(0)
    net = HipRandLANet(9, 6, num_neighbors=K, return_logits=True).to(device).eval()
    with torch.no_grad():
        a = net(x.to(device), posd, None, ptrd)
        dec = [i.clone() for i in net.last_decimation_idx]
        lvl = [N // 4 ** l for l in range(5)]
        dec1 = [d[1 * lvl[l + 1]:2 * lvl[l + 1]].long() - 1 * lvl[l] for l, d in enumerate(dec)]
        full = net(x.to(device), posd, None, ptrd, decimation_idx=dec)
        one = net(x[N:2 * N].to(device), posd[N:2 * N], None, torch.tensor([0, N], device=device), decimation_idx=dec1)
    assert a.shape == (B * N, 6) and bool(torch.isfinite(a).all())
    assert torch.allclose(full[N:2 * N], one, rtol=1e-4, atol=1e-4)  # tiles are independent units
    net.train()
    loss = myria3d_amd.cross_entropy(net(x.to(device), posd, None, ptrd), y.to(device), ignore_index=65)
    loss.backward()
    assert bool(torch.isfinite(loss)) and 0.5 < loss.item() < 5.0
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())


def _lookahead_pair(device):
    from myria3d_amd import HipRandLANet
    from oracle.randla_oracle import synthetic_batch

    x, pos, batch, ptr, _ = synthetic_batch([3000, 2500, 4100])
    x, pos, ptr = x.to(device), pos.to(device), ptr.to(device)
    torch.manual_seed(0)
    a = HipRandLANet(9, 6, return_logits=True).to(device).eval()
    torch.manual_seed(0)
    b = HipRandLANet(9, 6, return_logits=True).to(device).eval()
    b.load_state_dict(a.state_dict())
    return a, b, x, pos, ptr


def test_geometry_lookahead_matches_the_default_path(device):
    """prefetch_geometry(): tables built one step ahead (side stream, persistent double-buffered slots) must be the
    tables the default path builds in place — same device-side decimation seeds, same kernels — so eval-mode logits (no
    dropout; the decimation stays random) agree bit for bit, step after step."""
    a, b, x, pos, ptr = _lookahead_pair(device)
    plan_a, plan_b = a.plan_for(ptr), b.plan_for(ptr)
    b.prefetch_geometry(pos, ptr, plan_b)  # prime: every forward below consumes what was prefetched one step earlier
    for step in range(4):
        with torch.no_grad():
            out_a = a(x, pos, None, ptr, plan=plan_a)
            if step == 1:   # every call order: prefetch for the next step before / interleaved with / after the forward
                b.prefetch_geometry(pos, ptr, plan_b)
                out_b = b(x, pos, None, ptr, plan=plan_b)
            elif step == 2:
                b.prefetch_geometry(pos, ptr, plan_b, interleave=True)
                out_b = b(x, pos, None, ptr, plan=plan_b)
            else:
                out_b = b(x, pos, None, ptr, plan=plan_b)
                b.prefetch_geometry(pos, ptr, plan_b, after="forward_start")
            for da, db in zip(a.last_decimation_idx, b.last_decimation_idx):
                assert torch.equal(da, db), f"decimation of step {step}"
            assert torch.equal(out_a, out_b), f"logits of step {step}"
    b.join_geometry()
    torch.cuda.synchronize()
    # a batch nobody prefetched for falls back to the in-place path
    x2, pos2 = x[:3000].contiguous(), pos[:3000].contiguous()
    ptr2 = torch.tensor([0, 3000], device=device)
    with torch.no_grad():
        assert b(x2, pos2, None, ptr2).shape == (3000, 6)


def test_geometry_lookahead_under_hipgraph_replay(device):
    """The form bench.py times: two captured steps (one per buffer set) replayed in turn, each consuming the tables the
    previous replay prefetched; compared with the eager default path step by step."""
    a, b, x, pos, ptr = _lookahead_pair(device)
    plan_a, plan_b = a.plan_for(ptr), b.plan_for(ptr)
    out_b = torch.empty(x.shape[0], 6, device=device)

    def step_b():
        with torch.no_grad():
            b.prefetch_geometry(pos, ptr, plan_b, interleave=True)
            out_b.copy_(b(x, pos, None, ptr, plan=plan_b))
            b.join_geometry()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        b.prefetch_geometry(pos, ptr, plan_b)  # prime
        step_b()
        with torch.no_grad():
            ref = [a(x, pos, None, ptr, plan=plan_a).clone()]
        assert torch.equal(ref[0], out_b)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graphs = []
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            step_b()
        graphs.append(g)
    for step in range(5):
        graphs[step % 2].replay()
        with torch.no_grad():
            want = a(x, pos, None, ptr, plan=plan_a)
        torch.cuda.synchronize()
        assert torch.equal(want, out_b), f"replay {step}"


def test_seeded_decimation_is_reproducible(device):
    """the decimation stream follows torch's global seed (the reference draws torch.randperm from the global
    generator, pyg_randla_net.py:221): same seed -> same surviving points and bit-identical eval logits, run after run
    (which points survive must not depend on the arbitrary order of points inside a kNN grid cell); another seed ->
    another draw."""
    from myria3d_amd import HipRandLANet
    from oracle.randla_oracle import synthetic_batch

    x, pos, batch, ptr, _ = synthetic_batch([3000, 2500, 4100])
    x, pos, ptr = x.to(device), pos.to(device), ptr.to(device)
    outs, decs = [], []
    for seed in (0, 0, 1):
        torch.manual_seed(seed)
        net = HipRandLANet(9, 6, return_logits=True).to(device).eval()
        if outs:
            net.load_state_dict(first_state)
        else:
            first_state = {k: v.clone() for k, v in net.state_dict().items()}
        with torch.no_grad():
            o = [net(x, pos, None, ptr) for _ in range(2)]
        outs.append(o)
        decs.append([d.clone() for d in net.last_decimation_idx])
    for a, b in zip(decs[0], decs[1]):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], outs[0][1])          # the stream advances from call to call
    assert not torch.equal(decs[0][0], decs[2][0])          # another seed, another draw


def test_split_bf16_mode_meets_the_fp32_tolerances(device):
    """``matmul_precision = "bf16x3"`` (round 5 experiment): the attention GEMMs of the LFA layers with >= 64 channels as
    split-bf16 products on the matrix cores, everything else fp32 — the net must meet the tolerances of the fp32 contract:
    eval logits 1e-4 + 1e-4 |ref| at 2 x 12 800 points, and (train mode, fp64 oracle) train logits 1e-3, loss, EVERY
    parameter gradient at the fp32 contract's bound (_grad_table), running statistics."""
    from oracle.randla_oracle import fixed_decimation_indices, synthetic_batch

    ref, net = _pair(device, seed=7)
    x, pos, batch, ptr, y = synthetic_batch([12800, 12800])
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=3)
    args = (x.to(device), pos.to(device), None, ptr.to(device))
    ref.eval(), net.eval()
    with torch.no_grad():
        out_r = ref(x, pos, batch, ptr, decimation_idx=dec)
        out32 = net(*args, decimation_idx=dec)
        net.matmul_precision = "bf16x3"
        out3 = net(*args, decimation_idx=dec)
    assert not torch.equal(out32, out3), "the split-bf16 kernels really ran"
    _report("bf16x3.eval_logits", out3, out_r, EVAL_RTOL, EVAL_ATOL)
    _argmax_agreement("bf16x3.eval_logits", out3, out_r)
    print(f"[parity] bf16x3 vs fp32 kernels: max |d logit| = {(out3 - out32).abs().max().item():.3e}")
    _train_parity(device, x, pos, batch, ptr, y, seed=7, precision="bf16x3")


def test_bf16_mode_within_the_stated_tolerance(device):
    """SURVEY 8c / BASELINE config 2: with the matrix-bound layers on bf16 matrix cores, logits stay within 3e-2 of the
    fp32 oracle and the argmax agrees on >= 99 % of the points (two 12 800-point tiles, eval); a training step gives
    finite gradients close to the fp32 path's; torch.autocast(bfloat16) selects the same kernels."""
    from oracle.randla_oracle import fixed_decimation_indices, synthetic_batch

    ref, net = _pair(device, seed=7)
    x, pos, batch, ptr, y = synthetic_batch([12800, 12800])
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=3)
    args = (x.to(device), pos.to(device), None, ptr.to(device))
    ref.eval(), net.eval()
    with torch.no_grad():
        out_r = ref(x, pos, batch, ptr, decimation_idx=dec)
        out32 = net(*args, decimation_idx=dec)
        net.matmul_precision = "bf16"
        out16 = net(*args, decimation_idx=dec)
        net.matmul_precision = "fp32"
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out_ac = net(*args, decimation_idx=dec)
        again32 = net(*args, decimation_idx=dec)
    assert torch.equal(out32, again32) and not torch.equal(out32, out16)
    assert torch.equal(out16, out_ac)                      # autocast(bf16) == matmul_precision "bf16"
    err = (out16.cpu() - out_r).abs().max().item()
    agree = (out16.cpu().argmax(1) == out_r.argmax(1)).float().mean().item()
    print(f"[parity] bf16 eval logits: max abs err {err:.3e}, argmax agreement {agree:.4f}")
    assert err <= 3e-2 and agree >= 0.99
    # one training step in each precision: same loss to 1e-2, gradients finite and close in relative L2
    mask = torch.ones(25600, 32, device=device)
    grads = {}
    for prec in ("fp32", "bf16"):
        net.load_state_dict(ref.state_dict())
        net.train()
        net.zero_grad(set_to_none=True)
        net.matmul_precision = prec
        out = net(*args, decimation_idx=dec, dropout_mask=mask)
        loss = torch.nn.functional.cross_entropy(out, y.to(device))
        loss.backward()
        grads[prec] = (loss.item(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    assert abs(grads["fp32"][0] - grads["bf16"][0]) <= 1e-2
    worst, worst_k = 0.0, ""
    for k, g32 in grads["fp32"][1].items():
        g16 = grads["bf16"][1][k]
        assert bool(torch.isfinite(g16).all()), k
        den = g32.norm().item()
        if den > 1e-6 and (g16 - g32).norm().item() / den > worst:
            worst, worst_k = (g16 - g32).norm().item() / den, k
    print(f"[parity] bf16 vs fp32 parameter gradients: worst relative L2 {worst:.3e} ({worst_k})")
    # (a sanity bound, not the parity bar — that is the logits check above, SURVEY section 8c: the worst parameter is a
    # small-norm one whose bf16-vs-fp32 difference moves with the summation order of the GEMMs: 0.083 with one
    # accumulator per wave, 0.116 with K split over the four waves of a workgroup)
    assert worst <= 0.15, worst_k


def test_bf16_weight_gradients_on_a_flattened_net(device):
    """With flat gradient buffers the weight gradients are deferred and batched (``m3d_linear_wgrad_batch``); in bf16 mode
    the matrix-bound ones (deep layers) take bf16 operands there.  Same step as above on a flattened net: every parameter
    gradient stays within the same bound of the fp32 step's, and the deep weight gradients are NOT the fp32-GEMM ones of the
    un-flattened bf16 run (the bf16 kernel really ran)."""
    import myria3d_amd
    from oracle.randla_oracle import fixed_decimation_indices, synthetic_batch

    ref, _ = _pair(device, seed=7)
    x, pos, batch, ptr, y = synthetic_batch([12800, 12800])
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=3)
    args = (x.to(device), pos.to(device), None, ptr.to(device))
    mask = torch.ones(25600, 32, device=device)
    grads = {}
    for tag, prec, flat in (("fp32", "fp32", False), ("bf16", "bf16", False), ("bf16_flat", "bf16", True),
                            ("fp32_flat", "fp32", True)):
        net = myria3d_amd.HipRandLANet(9, 6, num_neighbors=16, return_logits=True)
        net.load_state_dict(ref.state_dict())
        net = net.to(device)
        if flat:
            net.flatten_parameters()
            myria3d_amd.FusedAdam(net, lr=1e-3)  # owns the stream / queue the deferred gradients go through
        net.train()
        net.matmul_precision = prec
        out = net(*args, decimation_idx=dec, dropout_mask=mask)
        torch.nn.functional.cross_entropy(out, y.to(device)).backward()
        torch.cuda.synchronize()
        grads[tag] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    rel = lambda a, b: (a - b).norm().item() / max(b.norm().item(), 1e-30)
    worst = max((rel(grads["bf16_flat"][k], g), k) for k, g in grads["fp32"].items() if g.norm().item() > 1e-6)
    print(f"[parity] flattened bf16 vs fp32 parameter gradients: worst relative L2 {worst[0]:.3e} ({worst[1]})")
    assert worst[0] <= 0.15, worst
    worst32 = max((rel(grads["fp32_flat"][k], g), k) for k, g in grads["fp32"].items() if g.norm().item() > 1e-6)
    print(f"[parity] flattened fp32 vs fp32 parameter gradients: worst relative L2 {worst32[0]:.3e} ({worst32[1]})")
    assert worst32[0] <= 2e-3, worst32
    k = "block4.mlp2.lins.0.weight"  # 3 200 x 256 -> 512: a 4 x 4-tile job of the batched launch
    assert not torch.equal(grads["bf16_flat"][k], grads["bf16"][k])
    assert rel(grads["bf16_flat"][k], grads["bf16"][k]) <= 2e-2


@pytest.mark.parametrize("sizes,k", [([1300, 900], 16), ([640, 350], 32)])
def test_eval_mode_forward_is_differentiable_like_the_reference(device, sizes, k):
    """The reference's eval-mode forward records an autograd graph like any torch module (BatchNorm on its running
    statistics, no dropout); ``HipRandLANet`` does too: gradients of a loss on the eval logits w.r.t. every parameter and
    w.r.t. the input features against the fp64 oracle (5e-3 relative L2), running statistics untouched."""
    from oracle.randla_oracle import fixed_decimation_indices

    ref, net = _pair(device, seed=9, k=k)
    ref = ref.double().eval()
    net.eval()
    x, pos, batch, ptr = rand_batch(sizes, seed=k)
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=5)
    y = torch.from_numpy(np.random.RandomState(3).randint(0, 6, (sum(sizes),)))
    xr = x.double().requires_grad_(True)
    out_r = ref(xr, pos.double(), batch, ptr, decimation_idx=dec)
    torch.nn.functional.cross_entropy(out_r, y).backward()
    before = {n_: b.clone() for n_, b in net.named_buffers()}
    xg = x.to(device).requires_grad_(True)
    out_g = net(xg, pos.to(device), None, ptr.to(device), decimation_idx=dec)
    assert out_g.requires_grad
    _report("eval_grad.logits", out_g, out_r, 2e-4, 2e-4)
    torch.nn.functional.cross_entropy(out_g, y.to(device)).backward()
    a, b = xg.grad.cpu().double(), xr.grad
    assert (a - b).norm().item() <= 5e-3 * b.norm().item(), "input gradient"
    worst = ("", 0.0)
    gr = dict(ref.named_parameters())
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        g0, g1 = p.grad.cpu().double(), gr[name].grad
        rel = (g0 - g1).norm().item() / max(g1.norm().item(), 1e-12)
        worst = max(worst, (name, rel), key=lambda t: t[1])
        assert rel <= 5e-3, (name, rel)
    print(f"[parity] eval-mode gradients, worst relative L2 error: {worst[0]} {worst[1]:.3e}")
    for n_, b_ in net.named_buffers():
        assert torch.equal(b_, before[n_]), n_  # eval mode: no statistic is updated
    with torch.no_grad():  # and the inference path (no graph) gives the same logits
        out_n = net(x.to(device), pos.to(device), None, ptr.to(device), decimation_idx=dec)
    assert not out_n.requires_grad and torch.allclose(out_n, out_g.detach(), rtol=1e-5, atol=1e-6)
