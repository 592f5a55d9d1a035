"""Where does the N > 1 launch form lose its 0.5 ms?  One GraphedStep training step (BASELINE config 2) timed in a fresh
process per situation: python tools/collective_probe.py CASE
  none      no process group (the N = 1 line)
  pg        a 1-rank RCCL process group exists, the optimizer does NOT exchange gradients
  pg_gloo   the same with a gloo group (torch.distributed without RCCL)
  eager     RCCL group, all-reduce + Adam after the graph (round 3's form)
  captured  RCCL group, all-reduce + Adam inside the step's graph
  noopt     no process group, optimizer step OUTSIDE the graph (graph B, then one eager Adam launch)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from myria3d_amd import FusedAdam, GraphedStep, HipRandLANet
from myria3d_amd.synthetic import synthetic_batch

case = sys.argv[1]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if case in ("pg", "eager", "captured"):
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29571", rank=0, world_size=1, device_id=dev)
elif case == "pg_gloo":
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29572", rank=0, world_size=1)
x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
x, pos, ptr, y = x.to(dev), pos.to(dev), ptr.to(dev), y.to(dev)
torch.manual_seed(0)
net = HipRandLANet(9, 6, decimation=4, num_neighbors=16, return_logits=True).to(dev)
net.flatten_parameters()
coll = case in ("eager", "captured")
opt = FusedAdam(net, lr=0.0039, all_reduce=coll, force_collective=coll)
kw = {}
if case == "noopt":
    kw["optimizer_in_graph"] = False
gs = GraphedStep(net, ptr, 9, mode="train", optimizer=opt, collective="eager" if case == "eager" else "captured", **kw)
gs.load_all(x, pos, y)
gs.prepare(preserve_state=False)
for _ in range(8):
    gs.step()
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        gs.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"collective_probe {case}: {min(ts):.3f} ms per step (runs {', '.join(f'{t:.3f}' for t in ts)}); collective={gs.collective} opt_in_graph={gs.opt_in_graph} side-stream candidates {gs.side_stream_ms}", flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
