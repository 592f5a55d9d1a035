"""host time of hipGraphLaunch vs GPU time of the captured training step"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
sys.argv = ["bench.py", "--steps", "5", "--warmup", "2", "--skip-cpu-baseline", "--skip-roofline", "--skip-extras"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
# reuse bench's builder but grab the graph: monkeypatch timed() to expose step_fn
holder = {}
orig = bench.timed
def spy(fn, steps, world):
    holder.setdefault("fns", []).append(fn)
    return orig(fn, steps, world)
bench.timed = spy
res = bench.train_bench(args, dev, 1, 0, args.tiles, args.points, args.neighbors, 5, 2)[0]
print("bench ms/step", res["ms_per_step"])
fn = holder["fns"][0]
for _ in range(5): fn()
torch.cuda.synchronize()
for n in (1, 4, 16, 64):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e3*(t1-t0)/n:.3f} ms/replay, total {1e3*(t2-t0)/n:.3f} ms/replay")
