"""Localise the step-3 NaN of the dual-graph GraphedStep: NaN checks on every state tensor after each step, for the
concurrent replay (A on its own stream) and a serialised one (A after B on the same stream)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests._util import fill_params_deterministic, rand_batch
from myria3d_amd import FusedAdam, GraphedStep, HipRandLANet
import numpy as np

dev = torch.device("cuda:0")
sizes = [int(v) for v in os.environ.get("SIZES", "2600,2200").split(",")]
xa, pa, _, ptr = rand_batch(sizes, seed=41)
xb, pb, _, _ = rand_batch(sizes, seed=42)
rs = np.random.RandomState(7)
ya = torch.from_numpy(rs.randint(0, 6, (sum(sizes),)))
yb = torch.from_numpy(rs.randint(0, 6, (sum(sizes),)))
to = lambda *ts: tuple(t.to(dev) for t in ts)
a, b = to(xa, pa, ya), to(xb, pb, yb)


def finite(name, t):
    if t.dtype in (torch.float32, torch.float64) and not bool(torch.isfinite(t).all()):
        print(f"    NON-FINITE: {name} ({int((~torch.isfinite(t)).sum())} of {t.numel()})")
        return False
    return True


for mode in os.environ.get("MODES", "concurrent").split(","):
    net = HipRandLANet(9, 6, return_logits=True)
    fill_params_deterministic(net, 31)
    net.mlp_classif.dropout = [0.0, 0.0]
    net = net.to(dev).flatten_parameters().train()
    opt = FusedAdam(net, lr=1e-3, eps=0.1)
    gs = GraphedStep(net, ptr, 9, mode="train", optimizer=opt)
    gs.load_all(*a); gs.load_next(*b)
    gs.prepare()
    net.set_decimation_seed(1234)
    if mode != "concurrent":
        gs._sA = torch.cuda.current_stream()  # A replays behind B on the same stream
    print(f"--- {mode}")
    for i in range(4):
        loss = gs.step()
        if mode == "serial_sync":
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        k = i & 1
        ok = finite("loss", loss)
        ok &= finite("params", net.flat_parameters) and finite("exp_avg", opt.exp_avg) and finite("exp_avg_sq", opt.exp_avg_sq)
        for kk in (0, 1):
            geo = net._look_slots[(id(gs), True, kk)].geo
            for nm, ts in (("mom", [m for m in geo.mom if m is not None]), ("pos4", geo.pos4)):
                for l, t in enumerate(ts):
                    ok &= finite(f"slot{kk}.{nm}[{l}]", t)
            for l, t in enumerate(geo.knn):
                bad = int(((t < -1) | (t >= geo.pos4[l].shape[0])).sum())
                if bad:
                    print(f"    slot{kk}.knn[{l}]: {bad} ids out of range"); ok = False
        badb = [n_ for n_, bf in net.named_buffers() if bf.dtype == torch.float32 and not bool(torch.isfinite(bf).all())]
        if badb:
            ok = False
            print(f"    {len(badb)} non-finite buffers, e.g. {badb[:6]}")
        enc = net.block1.lfa1.mlp_encoder.norms[0].module
        print(f"    block1.lfa1 enc running_var {enc.running_var.tolist()} running_mean {enc.running_mean.tolist()}")
        for kk in (0, 1):
            m0 = net._look_slots[(id(gs), True, kk)].geo.mom[0]
            print(f"    slot{kk}.mom[0][:4] = {m0[:4].tolist()}  sum|mom| = {float(m0.abs().sum()):.6e}  E = {gs.plan.num_edges[0]}")
        print(f"  step {i} (set {k}): loss {loss.item():.6f} {'ok' if ok else 'BAD'}")
        if i == 1:
            from myria3d_amd import ops
            geo = net._look_slots[(id(gs), True, 0)].geo
            for l in range(4):
                good = ops.lfa_moments(geo.pos4[l], geo.knn[l])
                torch.cuda.synchronize()
                d = (geo.mom[l] - good).abs()
                bad = (d > 1e-6 * good.abs().clamp(min=1.0)).nonzero().flatten().tolist()
                print(f"    slot0.mom[{l}]: {len(bad)} bad entries {bad[:12]}; slot-good at those: {[float(geo.mom[l][j] - good[j]) for j in bad[:6]]}")
                print(f"       ptr {geo.mom[l].data_ptr():#x}  neighbours: knn[{l}] {geo.knn[l].data_ptr():#x}+{geo.knn[l].numel()*4:#x}")
