"""eager training step time in blocks of 10 steps (does the eager path need a long warm-up?)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
sys.argv = ["bench.py", "--no-graph", "--steps", "10", "--warmup", "0", "--skip-cpu-baseline", "--skip-roofline", "--skip-extras"]
args = bench.parse()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
holder = []
orig = bench.timed
def spy(fn, steps, world):
    holder.append(fn)
    return orig(fn, steps, world)
bench.timed = spy
bench.train_bench(args, dev, 1, 0, args.tiles, args.points, args.neighbors, 10, 0)
fn = holder[0]
for blk in range(8):
    print(f"block {blk}: {orig(fn, 10, 1) / 10 * 1e3:.3f} ms/step")
