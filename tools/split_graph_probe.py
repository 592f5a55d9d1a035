"""GPU box: the step graph (B) cut into forward + loss (B1) and backward + Adam (B2), so that the position-only graph of the
next step (A) can start BETWEEN them instead of with B1 (external event-record nodes are not available on ROCm: the cut
is the only place an ordinary event can sit).  Prints ms per step for A started with B1 / with B2.
Round 3 (one box): B alone 4.446 ms, A alone 0.690 ms, GraphedStep 4.700 ms; B1 | B2 with A started with B1 4.706 ms, with
B2 (after the forward) 4.710 ms — the 0.25 ms the overlap costs does not depend on which half of the step A runs beside.
usage: python tools/split_graph_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import FusedAdam, GraphedStep, HipRandLANet, cross_entropy
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
t0 = time.perf_counter()
for _ in range(50):
    gs.step()
torch.cuda.synchronize()
print(f"GraphedStep (one graph B, A starts with it): {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per step")

# the same step as B1 | B2
net._finish_interleaved(); net._look_queue.clear()
gs._geo(0)
torch.cuda.synchronize()
g1, g2, gA = [], [], []
for k in range(2):
    s = gs.sets[k]
    a = torch.cuda.CUDAGraph()
    with torch.cuda.graph(a, capture_error_mode="thread_local"):
        out = net(s.x, s.pos, None, gs.ptr, plan=gs.plan)
        loss = cross_entropy(out, s.y, ignore_index=65)
    b = torch.cuda.CUDAGraph()
    with torch.cuda.graph(b, pool=a.pool(), capture_error_mode="thread_local"):
        loss.backward()
        net.grad_side.join()
        net.join_geometry()
        opt.step()
    c = torch.cuda.CUDAGraph()
    with torch.cuda.graph(c, stream=net._side_stream(dev), capture_error_mode="thread_local"):
        gs._geo(k ^ 1)
    g1.append(a), g2.append(b), gA.append(c)
    del out, loss
net._finish_interleaved(); net._look_queue.clear()
sA = torch.cuda.Stream()
evA, evReady, evMid = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()


def run(tag, a_with):
    gs._geo(0)  # tables for the first step
    torch.cuda.synchronize()
    evA.record(); evReady.record()
    turn = 0

    def step():
        nonlocal turn
        k = turn & 1
        cur = torch.cuda.current_stream()
        cur.wait_event(evA)
        sA.wait_event(evReady)
        g1[k].replay()
        if a_with == "B2":
            evMid.record(cur)
            sA.wait_event(evMid)
        g2[k].replay()
        with torch.cuda.stream(sA):
            gA[k].replay()
            evA.record(sA)
        evReady.record(cur)
        turn += 1

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per step")


run("B1 | B2, A starts with B1", "B1")
run("B1 | B2, A starts with B2 (after the forward)", "B2")
run("B1 | B2, A starts with B1 (again)", "B1")
