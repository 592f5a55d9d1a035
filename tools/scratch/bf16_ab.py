"""bf16-mode training step under the current environment knobs (A/B helper)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
sys.argv = ["bench.py", "--steps", "40", "--warmup", "10", "--launch", "graph", "--skip-cpu-baseline", "--skip-roofline", "--skip-extras"]
args = bench.parse()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
res = bench.train_bench(args, dev, 1, 0, args.tiles, args.points, args.neighbors, 40, 10, precision="bf16")[0]
print("bf16", os.environ.get("TAG", ""), res["ms_per_step"], res["fwd_only"]["ms_per_step"])
