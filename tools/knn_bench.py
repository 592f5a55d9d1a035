"""kNN timing at the BASELINE config-2 level shapes (hipGraph replay, median): python tools/knn_bench.py [pmc]
``pmc``: just launch the level-1 query a few times (target of a rocprofv3 --pmc pass)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import ops
from myria3d_amd.synthetic import synthetic_batch


def timeit(fn, reps=10, inner=10):
    """median over `reps` replays of a hipGraph holding `inner` back-to-back calls (no host launch overhead)"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner): fn()
    g.replay(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); g.replay(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3 / inner  # us per call (median)


dev = torch.device("cuda:0")
x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
pos, ptr = pos.to(dev), ptr.to(dev)
p4, ptrs = [ops.pad_pos(pos)], [ptr]
g = torch.Generator(device=dev).manual_seed(0)
for l in range(3):
    per = int(ptrs[-1][1].item())
    idx = torch.cat([b * per + torch.randperm(per, device=dev, generator=g)[: per // 4] for b in range(16)]).to(torch.int32)
    p4.append(ops.gather_rows(p4[-1], idx)); ptrs.append(ptrs[-1] // 4)
if "pmc" in sys.argv:
    index = ops.KnnIndex(p4[0], ptrs[0])
    for _ in range(4):
        index.query(16, qry=index, sorted_io=True)
    torch.cuda.synchronize()
    sys.exit(0)
out, nn = [], []
indices = [ops.KnnIndex(p4[l], ptrs[l]) for l in range(4)]
for l in range(4):
    index = indices[l]
    out.append(timeit(lambda: index.query(16, qry=index, sorted_io=True)))
for l in range(3):  # decoder 1-NN tables: every level-l point among the level l+1 points
    src, qry = indices[l + 1], indices[l]
    nn.append(timeit(lambda: src.query(1, qry=qry, sorted_io=True)))
tag = os.environ.get("M3D_LIB", "default").split("libm3d_")[-1]
direct = [timeit(lambda: indices[l].query(16, qry=indices[l], sorted_io=True, kernel="direct")) for l in range(2)]
print(f"knn_bench lib={tag}: " + " ".join(f"L{l+1}={t:.1f}us" for l, t in enumerate(out)) + " | 1-NN " + " ".join(f"{t:.1f}" for t in nn)
      + " | direct-insertion kernel L1/L2 " + " ".join(f"{t:.1f}" for t in direct))
