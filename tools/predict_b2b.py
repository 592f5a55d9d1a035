"""GPU box: predict_cloud on bench.py's 10 M-point cloud as clouds back to back (no synchronisation between calls), per group of
K calls: is the cost of back-to-back clouds (52.5 vs 48.8 ms) a steady state or the allocator's pool still growing?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myria3d_amd import predict_cloud
dev = torch.device("cuda:0")
net, pos, x = bench.predict_e2e_inputs(dev)
run = lambda: predict_cloud(net, pos, x, tile_width=1000.0, subtile_width=50, batch_size=50)
run(); torch.cuda.synchronize()
for group in range(5):
    k = 2 if group < 3 else 6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3 / k
    st = torch.cuda.memory_stats()
    print(f"group {group}: {k} clouds back to back: {dt:.2f} ms per cloud; reserved {st['reserved_bytes.all.current'] / 2**30:.2f} GiB, "
          f"hipMalloc calls so far {st['num_device_alloc']}")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    print(f"one cloud, synchronised: {(time.perf_counter() - t0) * 1e3:.2f} ms; hipMalloc calls {torch.cuda.memory_stats()['num_device_alloc']}")
