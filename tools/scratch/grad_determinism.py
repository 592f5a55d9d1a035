"""run-to-run and flat-vs-plain differences of every parameter gradient of one fp32 training step (2 x 12 800 points)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import myria3d_amd
from oracle.randla_oracle import RandLANetOracle, fixed_decimation_indices, synthetic_batch
from tests._util import fill_params_deterministic
dev = torch.device("cuda:0")
ref = RandLANetOracle(9, 6, return_logits=True); fill_params_deterministic(ref, 7)
x, pos, batch, ptr, y = synthetic_batch([12800, 12800])
dec = fixed_decimation_indices(ptr.tolist(), 4, seed=3)
args = (x.to(dev), pos.to(dev), None, ptr.to(dev))
mask = torch.ones(25600, 32, device=dev)
def run(flat):
    net = myria3d_amd.HipRandLANet(9, 6, num_neighbors=16, return_logits=True)
    net.load_state_dict(ref.state_dict()); net = net.to(dev)
    if flat:
        net.flatten_parameters(); myria3d_amd.FusedAdam(net, lr=1e-3)
    net.train()
    rec = {}
    out = net(*args, decimation_idx=dec, dropout_mask=mask, record=rec)
    torch.nn.functional.cross_entropy(out, y.to(dev)).backward()
    torch.cuda.synchronize()
    return out.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}
rel = lambda a, b: (a - b).norm().item() / max(b.norm().item(), 1e-30)
runs = [run(False), run(False), run(True), run(True)]
names = ["plain#0", "plain#1", "flat#0", "flat#1"]
for i in range(1, 4):
    worst = max((rel(runs[i][1][k], g), k) for k, g in runs[0][1].items() if g.norm().item() > 1e-6)
    print(f"{names[i]} vs {names[0]}: logits max |d| {(runs[i][0] - runs[0][0]).abs().max().item():.3e}; worst grad rel-L2 {worst[0]:.3e} ({worst[1]})")
