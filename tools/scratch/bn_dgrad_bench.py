"""fused BatchNorm-backward + dgrad (ops.bn_dgrad) vs the two-pass path, level-1 layer shapes (hipGraph timing)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myria3d_amd import ops

def timeit(fn, reps=10, inner=10):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner): fn()
    g.replay(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); g.replay(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3 / inner

dev = torch.device("cuda:0")
for M, N, Kin in ((204800, 4, 32), (204800, 8, 8), (204800, 16, 16), (204800, 32, 16), (204800, 32, 32), (204800, 64, 32),
                  (204800, 32, 64), (51200, 16, 32), (51200, 64, 64), (51200, 128, 64), (12800, 256, 128), (3200, 512, 256)):
    dy = torch.randn(M, N, device=dev); z = torch.randn(M, N, device=dev)
    w = torch.randn(N, Kin, device=dev) / N ** 0.5
    sc = torch.rand(N, device=dev) + 0.5; sh = torch.randn(N, device=dev) * 0.1
    mu = torch.randn(N, device=dev) * 0.1; isd = torch.rand(N, device=dev) + 0.5
    t_f = timeit(lambda: ops.bn_dgrad(dy, z, sc, sh, mu, isd, True, w))
    def two():
        dz = ops.bn_bwd(dy, z, sc, sh, mu, isd, True)[0]
        return ops.linear_dgrad(dz, w)
    t_2 = timeit(two)
    mb = M * (2 * N + N + Kin) * 4 / 1e6
    print(f"bn_dgrad M={M:6d} N={N:3d} Kin={Kin:3d}: fused {t_f:6.1f} us   two-pass {t_2:6.1f} us   ({mb:6.1f} MB -> {mb / t_f / 1e3 * 1e3:5.0f} GB/s fused)")
