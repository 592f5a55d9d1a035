"""Launches the kernels of bench.py's roofline entries (and nothing else heavy) — the target of the rocprofv3 --pmc
passes whose per-kernel FETCH_SIZE / WRITE_SIZE averages land in profiles/ (tools/gpu_pmc.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myria3d_amd import HipRandLANet, make_plan
from myria3d_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
pos, ptr = pos.to(dev), ptr.to(dev)
torch.manual_seed(0)
net = HipRandLANet(9, 6, return_logits=True).to(dev)
plan = make_plan(ptr.tolist(), 4, 16, dev)
print(bench.stage_rooflines(net, pos, plan))
