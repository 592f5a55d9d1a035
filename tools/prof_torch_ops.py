"""Which torch-level ops (copies, fills, adds ...) still run inside one training step? (GPU box, via gpurun)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import FusedAdam, HipRandLANet, cross_entropy, make_plan
from myria3d_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
x, pos, batch, ptr, y = synthetic_batch([12800] * 4)
x, pos, ptr, y = x.to(dev), pos.to(dev), ptr.to(dev), y.to(dev)
net = HipRandLANet(9, 6, return_logits=True).to(dev).flatten_parameters()
opt = FusedAdam(net, lr=1e-3)
plan = make_plan(ptr.tolist(), 4, 16, dev)
def step():
    net.train()
    loss = cross_entropy(net(x, pos, None, ptr, plan=plan), y, 65)
    loss.backward(); opt.step()
step(); step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
import collections
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::mul", "aten::div"):
        st = [f for f in ev.stack if "myria3d_amd" in f or "tools/" in f or "autograd" in f][:2]
        cnt[(ev.name, " <- ".join(st))] += 1
for (name, st), n in cnt.most_common(40):
    print(f"{n:4d} {name:14s} {st}")
