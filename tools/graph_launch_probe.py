"""GPU box: how long the HOST spends inside the two hipGraphLaunch calls of a GraphedStep training step.
Round 3: B (the step, ~180 kernel nodes) 114-133 us, A (position-only, ~35 nodes) 21-23 us — the launch is not what
makes A start late in the rocprofv3 timelines (that is the tracer); replaying A before B measured 4.697 vs 4.687 ms.
usage: python tools/graph_launch_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import FusedAdam, GraphedStep, HipRandLANet
from myria3d_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
x, pos, ptr, y = x.to(dev), pos.to(dev), ptr.to(dev), y.to(dev)
torch.manual_seed(0)
net = HipRandLANet(9, 6, decimation=4, num_neighbors=16, return_logits=True).to(dev)
net.flatten_parameters()
opt = FusedAdam(net, lr=3.9e-3)
gs = GraphedStep(net, ptr, 9, mode="train", optimizer=opt, ignore_index=65, lookahead=True, launch="graph")
gs.load_all(x, pos, y)
gs.prepare(preserve_state=False)
for _ in range(10):
    gs.step()
torch.cuda.synchronize()
gB, gA = gs._graphs
# host time of the bare replay calls (GPU idle before each: the call itself, not back-pressure)
for name, g in (("B (step)", gB[0]), ("A (position-only)", gA[0])):
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()
        ts.append((t1 - t0) * 1e6)
    torch.cuda.synchronize()
    print(f"host time of hipGraphLaunch {name}: median {sorted(ts)[5]:.0f} us (min {min(ts):.0f})")
def run(tag):
    gs.prime()
    for _ in range(10):
        gs.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        gs.step()
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per step")


run("default stream")
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range", lo, hi)
hp = torch.cuda.Stream(priority=-1)
hp.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(hp):  # the step's graph replayed on a high-priority stream, the position-only graph on a normal one
    run("step graph on a high-priority stream")
torch.cuda.current_stream().wait_stream(hp)
run("default stream again")
