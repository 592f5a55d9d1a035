"""GPU box: the partial-sum reduce of the LFA backward (m3d_lfa_bwd_reduce_batch) alone, per layer shape of BASELINE config 2
and for all eight layers in one launch (what the step does).  usage: [M3D_LIB=...] python tools/lfa_reduce_bench.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import ops
from myria3d_amd._lib import lib

dev = torch.device("cuda:0")
LAYERS = [(204800, 8), (204800, 16), (51200, 32), (51200, 64), (12800, 64), (12800, 128), (3200, 128), (3200, 256)]
K = 16


def timeit(fn, reps=10, inner=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]


def job(n, ch):
    nb = lib().m3d_lfa_bwd_workspace_bytes(n, K, ch)
    ws = torch.randn(nb // 4 + 64, device=dev)
    return n, ch, ws, torch.zeros(ch, ch, device=dev), torch.zeros(11 * (ch // 2), dtype=torch.float64, device=dev), nb


def run(jobs):
    m = len(jobs)
    vp = lambda k: (ctypes.c_void_p * m)(*[j[k].data_ptr() for j in jobs])
    ops.call("m3d_lfa_bwd_reduce_batch", m, (ctypes.c_int64 * m)(*[j[0] for j in jobs]), (ctypes.c_int32 * m)(*[K] * m),
             (ctypes.c_int32 * m)(*[j[1] for j in jobs]), vp(2), vp(3), vp(4), ops._st())


jobs = [job(n, ch) for n, ch in LAYERS]
tot = 0
for j in jobs:
    t = timeit(lambda: run([j]))
    tot += j[5]
    print(f"n={j[0]:7d} ch={j[1]:4d}  partials {j[5] / 1e6:7.1f} MB  {t:7.1f} us  ({j[5] / t / 1e6:6.2f} TB/s)")
t = timeit(lambda: run(jobs))
print(f"all eight layers in one launch: {tot / 1e6:7.1f} MB  {t:7.1f} us  ({tot / t / 1e6:6.2f} TB/s)")
