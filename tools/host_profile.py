"""Where the HOST time of an eagerly launched training step goes (cProfile over 20 steps): python tools/host_profile.py"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import HipRandLANet, make_plan
from myria3d_amd.synthetic import synthetic_batch
from myria3d_amd.train import FusedAdam, cross_entropy

dev = torch.device("cuda:0")
x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
x, pos, ptr, y = x.to(dev), pos.to(dev), ptr.to(dev), y.to(dev)
torch.manual_seed(0)
net = HipRandLANet(9, 6, return_logits=True).to(dev)
net.flatten_parameters()
plan = make_plan(ptr.tolist(), 4, 16, dev)
opt = FusedAdam(net, lr=0.004)


def step():
    net.train()
    net.prefetch_geometry(pos, ptr, plan, train=True)
    out = net(x, pos, None, ptr, plan=plan)
    loss = cross_entropy(out, y, ignore_index=65)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host {1e3 * (t1 - t0) / 20:.3f} ms/step, wall {1e3 * (t2 - t0) / 20:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
