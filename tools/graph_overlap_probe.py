"""GPU box: WHEN do the two graphs of a GraphedStep step run relative to each other, without a tracer (HIP events on the
two replay streams)?  B = the step (forward, backward, Adam) on the caller's stream, A = the position-only work for the NEXT
step on the side stream.  Prints, per step: B's duration, A's start / end relative to B's start, and how long the next B
waits after this B ended (what of A is NOT hidden).
usage: python tools/graph_overlap_probe.py"""
import os, sys, torch
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
cur, sA = torch.cuda.current_stream(), gs._sA
N = 12
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N)]
for i in range(N):
    k = gs.turn % 2
    cur.wait_event(gs._evA)
    sA.wait_event(gs._evReady)
    ev[i][0].record(cur)
    gB[k].replay()
    ev[i][1].record(cur)
    with torch.cuda.stream(sA):
        ev[i][2].record(sA)
        gA[k].replay()
        ev[i][3].record(sA)
        gs._evA.record(sA)
    gs._evReady.record(cur)
    gs.turn += 1
torch.cuda.synchronize()
for i in range(2, N - 1):
    b0, b1, a0, a1 = ev[i]
    print(f"step {i}: B {b0.elapsed_time(b1) * 1e3:7.1f} us | A starts {b0.elapsed_time(a0) * 1e3:7.1f}, ends "
          f"{b0.elapsed_time(a1) * 1e3:7.1f} after B's start | next B starts {b1.elapsed_time(ev[i + 1][0]) * 1e3:6.1f} us after this B ended"
          f" | step period {b0.elapsed_time(ev[i + 1][0]) * 1e3:7.1f} us")
