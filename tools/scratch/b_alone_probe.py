"""GPU box: what the position-only graph (A: the NEXT step's kNN tables, decimation, reverse lists) costs the step it runs beside.
Times (HIP events, 200 replays each) on bench.py's batch: the normal dual-graph step; graph B alone (the tables of its slot
stay as they are: same batch); graph A alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myria3d_amd import FusedAdam, GraphedStep, HipRandLANet
from myria3d_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
x, pos, ptr, y = x.to(dev), pos.to(dev), ptr.to(dev), y.to(dev)
torch.manual_seed(0)
net = HipRandLANet(9, 6, decimation=4, num_neighbors=16, return_logits=True).to(dev)
net.flatten_parameters()
opt = FusedAdam(net, lr=0.0039, all_reduce=True)
gs = GraphedStep(net, ptr, x.shape[1], mode="train", optimizer=opt, ignore_index=65, lookahead=True, launch="graph", lookahead_mode="dual")
gs.load_all(x, pos, y)
for _ in range(20):
    gs.step()
torch.cuda.synchronize()

def timed(fn, n=200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

t_step = timed(gs.step)
gB, gA = gs._graphs
torch.cuda.synchronize()
t_b = timed(lambda: gB[0].replay())
sA = gs._sA
def a_only():
    with torch.cuda.stream(sA):
        gA[0].replay()
cur = torch.cuda.current_stream()
torch.cuda.synchronize()
ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(sA):
    ea.record(sA)
    for _ in range(200):
        gA[0].replay()
    eb.record(sA)
torch.cuda.synchronize()
t_a = ea.elapsed_time(eb) / 200
print(f"dual-graph step {t_step:.4f} ms | graph B alone {t_b:.4f} ms | graph A alone {t_a:.4f} ms | B + A serial {t_b + t_a:.4f} | overlap hides {t_b + t_a - t_step:.4f} of A's {t_a:.4f}")
