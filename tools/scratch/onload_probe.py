import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests._util import fill_params_deterministic, rand_batch
from myria3d_amd import FusedAdam, HipRandLANet, cross_entropy, make_plan, ops
from oracle.randla_oracle import fixed_decimation_indices
device = torch.device("cuda:0")
sizes = [700, 500]
x, pos, batch, ptr = rand_batch(sizes, seed=13)
dec = [d.to(device) for d in fixed_decimation_indices(ptr.tolist(), 4, seed=2)]
rs = np.random.RandomState(6)
mask = torch.from_numpy((rs.uniform(size=(sum(sizes), 32)) > 0.5).astype(np.float32)).to(device)
y = torch.from_numpy(rs.randint(0, 6, (sum(sizes),))).to(device)
xd, pd, ptrd = x.to(device), pos.to(device), ptr.to(device)
plan = make_plan(ptr.tolist(), 4, 16, device)
def run(onload, steps=1):
    ops.BN_ON_LOAD = onload
    net = HipRandLANet(9, 6, return_logits=True)
    fill_params_deterministic(net, 31)
    net = net.to(device).flatten_parameters().train()
    opt = FusedAdam(net, lr=1e-3)
    outs = []
    for _ in range(steps):
        out = net(xd, pd, None, ptrd, decimation_idx=dec, dropout_mask=mask, plan=plan)
        loss = cross_entropy(out, y, 65)
        loss.backward()
        if net.grad_side is not None: net.grad_side.join()
        torch.cuda.synchronize()
        outs.append((out.detach().clone(), net.flat_grads.clone(), {k: v.clone() for k, v in net.state_dict().items() if "running" in k}))
        opt.step()
    return outs
a = run(True, 2); b = run(True, 2); c = run(False, 2)
for s in range(2):
    print("step", s, "on-load vs on-load: logits", (a[s][0] - b[s][0]).abs().max().item(), "grads", (a[s][1] - b[s][1]).abs().max().item())
    print("step", s, "on-load vs separate: logits", (a[s][0] - c[s][0]).abs().max().item(), "grads", (a[s][1] - c[s][1]).abs().max().item(),
          "rel", ((a[s][1] - c[s][1]).norm() / c[s][1].norm()).item())
    worst = max(((a[s][2][k] - c[s][2][k]).abs().max().item(), k) for k in a[s][2])
    print("        running stats worst", worst)
