"""GPU box: would two same-shape deep GEMMs share the chip?  time(M) vs time(2M) for the residual-tail shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "none"]
from myria3d_amd import ops
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opbench.py")).read().split("LEVELS =")[0])
for M, K, N in ((3200, 256, 512), (6400, 256, 512), (12800, 128, 256), (25600, 128, 256), (51200, 64, 128), (102400, 64, 128)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    dz = torch.randn(M, N, device=dev)
    st = ops.stat_slots(N, dev, M)
    t1 = timeit(lambda: ops.gemm(x, w, M, N, K, bias=b, stats=st, stat_slots=True))
    t2 = timeit(lambda: ops.linear_dgrad(dz, w))
    print(f"M={M:6d} K={K:4d} N={N:4d}: fwd+stats {t1:6.1f} us  dgrad {t2:6.1f} us")
