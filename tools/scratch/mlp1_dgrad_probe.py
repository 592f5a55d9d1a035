"""GPU box: why does block1.mlp1's fused dz + dgrad launch take 89 us inside the step (profiles/r05mid_step_timeline.csv,
r04zzzzz likewise) and 16 us alone?  Times m3d_bn_dgrad_f32 at its shape (M = 204 800, N = 4, Kin = 32) with / without the
accumulate flag and gradient sinks, alone and behind the kernels that precede it in the step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myria3d_amd import ops
dev = torch.device("cuda:0")
M, N, K = 204800, 4, 32
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3
ops.arena.stop()
w = torch.randn(N, K, device=dev)
z = torch.randn(M, N, device=dev)
sc, sh, mu, isd = (torch.rand(N, device=dev) + 0.5 for _ in range(4))
acc = torch.randn(M, K, device=dev)
dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
for name, dy in (("random dy", torch.randn(M, N, device=dev)), ("tiny dy (1e-6)", torch.randn(M, N, device=dev) * 1e-6),
                 ("denormal dy (1e-40)", torch.full((M, N), 1e-40, device=dev)), ("zero dy", torch.zeros(M, N, device=dev))):
    print(name,
          "plain:", round(timeit(lambda: ops.bn_dgrad(dy, z, sc, sh, mu, isd, True, w)), 1),
          "acc:", round(timeit(lambda: ops.bn_dgrad(dy, z, sc, sh, mu, isd, True, w, acc=acc)), 1),
          "acc+sinks:", round(timeit(lambda: ops.bn_dgrad(dy, z, sc, sh, mu, isd, True, w, sinks=(dg, db), acc=acc)), 1), "us")
accd = torch.full((M, K), 1e-41, device=dev)
dy = torch.randn(M, N, device=dev)
print("denormal accumulator:", round(timeit(lambda: ops.bn_dgrad(dy, z, sc, sh, mu, isd, True, w, acc=accd.clone())), 1), "us (incl. clone)")
