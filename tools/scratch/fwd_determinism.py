"""which recorded tensor of the train-mode forward differs first between two identical runs"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import myria3d_amd
from myria3d_amd import ops
from oracle.randla_oracle import RandLANetOracle, fixed_decimation_indices, synthetic_batch
from tests._util import fill_params_deterministic
dev = torch.device("cuda:0")
ref = RandLANetOracle(9, 6, return_logits=True); fill_params_deterministic(ref, 7)
x, pos, batch, ptr, y = synthetic_batch([12800, 12800])
dec = fixed_decimation_indices(ptr.tolist(), 4, seed=3)
args = (x.to(dev), pos.to(dev), None, ptr.to(dev))
mask = torch.ones(25600, 32, device=dev)
recs = []
for rep in range(2):
    net = myria3d_amd.HipRandLANet(9, 6, num_neighbors=16, return_logits=True)
    net.load_state_dict(ref.state_dict()); net = net.to(dev).train()
    rec = {}
    with torch.no_grad():
        out = net(*args, decimation_idx=dec, dropout_mask=mask, record=rec)
    rec["logits"] = out
    recs.append({k: v.detach().clone() for k, v in rec.items() if torch.is_tensor(v)})
    bufs = {k: v.detach().clone() for k, v in net.named_buffers() if "running" in k}
    recs[-1].update(bufs)
for k in recs[0]:
    a, b = recs[0][k].double(), recs[1][k].double()
    d = (a - b).abs().max().item()
    if d > 0:
        print(f"{k:50s} max |d| {d:.3e}  (|ref| max {a.abs().max().item():.3e})")
print("done")
# the first SharedMLP layer on its own: fc0 output -> mlp1 of block 1 (32 -> 4), shortcut (32 -> 32)
torch.manual_seed(1)
h = torch.randn(25600, 32, device=dev)
for N in (4, 32, 8, 16):
    w = torch.randn(N, 32, device=dev) * 0.2; b = torch.randn(N, device=dev)
    outs = []
    for rep in range(3):
        st = torch.zeros((ops.bn_slots(25600), 2, N), dtype=torch.float64, device=dev)
        z = ops.gemm(h, w, 25600, N, 32, bias=b, stats=st, stat_slots=True)
        outs.append((z.clone(), st.clone()))
    print(N, [torch.equal(outs[0][0], o[0]) for o in outs[1:]], [(outs[0][1].sum(0) - o[1].sum(0)).abs().max().item() for o in outs[1:]],
          "slots", st.shape[0])
