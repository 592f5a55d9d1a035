import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myria3d_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for M, K, N in ((2200, 32, 4), (2200, 32, 32), (2200, 8, 8), (2200, 16, 16), (550, 32, 16), (550, 64, 128), (137, 128, 256), (34, 256, 512), (8, 512, 512),
                (204800, 32, 32), (3200, 256, 512), (12800, 128, 128)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    res = []
    for rep in range(4):
        st = torch.zeros((ops.bn_slots(M), 2, N), dtype=torch.float64, device=dev)
        z = ops.gemm(x, w, M, N, K, bias=b, stats=st, stat_slots=True)
        res.append((z.clone(), st.sum(0).clone()))
    zd = res[0][0].double()
    ref = torch.stack([zd.sum(0), (zd * zd).sum(0)])
    same_z = all(torch.equal(res[0][0], r[0]) for r in res[1:])
    dev_st = max((res[0][1] - r[1]).abs().max().item() for r in res[1:])
    err = ((res[0][1] - ref).abs() / (ref.abs() + 1e-3)).max().item()
    print(f"M={M:6d} K={K:3d} N={N:3d}: z identical {same_z}, stats run-to-run max |d| {dev_st:.3e}, vs fp64 of z rel {err:.3e}")
