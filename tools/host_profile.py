"""Where the HOST time of an eagerly launched training step goes (cProfile over 20 steps): python tools/host_profile.py [variable]
``variable``: a different tile layout every step (the opt-in path of INTEGRATION.md section 3: flat buffers, FusedAdam, HIP
criterion, the next batch's position-only work interleaved) — what a loop with per-batch layouts pays on the host."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import HipRandLANet, make_plan
from myria3d_amd.synthetic import synthetic_batch
from myria3d_amd.train import FusedAdam, cross_entropy

if "st" in sys.argv:  # backward nodes run on the calling thread (no hand-off to the autograd device thread per node)
    torch.autograd.set_multithreading_enabled(False)
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


if "variable" in sys.argv:
    import numpy as np

    rs = np.random.RandomState(0)
    var = []
    for i in range(8):
        sizes = [int(v) for v in rs.randint(6400, 19201, size=16)]
        vx, vpos, vbatch, vptr, vy = synthetic_batch(sizes, first_tile_id=100 * i)
        var.append(tuple(t.to(dev) for t in (vx, vpos, vbatch, vptr, vy)))
    turn = [0]
    net.train()

    def step():  # noqa: F811
        vx, vpos, vbatch, vptr, vy = var[turn[0] % len(var)]
        nxt = var[(turn[0] + 1) % len(var)]
        turn[0] += 1
        net.prefetch_geometry(nxt[1], nxt[3], interleave=True)
        cross_entropy(net(vx, vpos, vbatch, vptr), vy, ignore_index=65).backward()
        opt.step()

    net.prefetch_geometry(var[0][1], var[0][3])


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
st.sort_stats("tottime").print_stats(40)
