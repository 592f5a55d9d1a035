"""Per-op timing at the BASELINE config-2 shapes (16 x 12 800 points): run on the GPU box via gpurun.
usage: python tools/opbench.py [gemm] [wgrad] [knn] [lfa] [bn] [bnbwd]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import ops
dev = torch.device("cuda:0")
FULL = "nofull" not in sys.argv  # lfa: complete neighbourhoods promised (the mask-free kernels of round 5); "nofull": the general ones
what = (set(sys.argv[1:]) - {"nofull"}) or {"gemm", "wgrad", "knn", "lfa", "bn"}

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

LEVELS = [204800, 51200, 12800, 3200, 800]
# (name, M, K(k0,k1), N)
FWD = [("fc0", 0, 9, 0, 32), ("b1.mlp1", 0, 32, 0, 4), ("b1.post1", 0, 8, 0, 8), ("b1.post2", 0, 16, 0, 16), ("b1.mlp2", 0, 16, 0, 32),
       ("b1.short", 0, 32, 0, 32), ("fp1", 0, 32, 32, 32), ("cls1", 0, 32, 0, 64), ("cls2", 0, 64, 0, 32), ("fc_cls", 0, 32, 0, 6),
       ("b2.mlp1", 1, 32, 0, 16), ("b2.post1", 1, 32, 0, 32), ("b2.post2", 1, 64, 0, 64), ("b2.mlp2", 1, 64, 0, 128), ("b2.short", 1, 32, 0, 128), ("fp2", 1, 128, 32, 32),
       ("b3.mlp1", 2, 128, 0, 32), ("b3.post1", 2, 64, 0, 64), ("b3.post2", 2, 128, 0, 128), ("b3.mlp2", 2, 128, 0, 256), ("b3.short", 2, 128, 0, 256), ("fp3", 2, 256, 128, 128),
       ("b4.mlp1", 3, 256, 0, 64), ("b4.post1", 3, 128, 0, 128), ("b4.post2", 3, 256, 0, 256), ("b4.mlp2", 3, 256, 0, 512), ("b4.short", 3, 256, 0, 512), ("fp4", 3, 512, 256, 256),
       ("summit", 4, 512, 0, 512)]
if "gemm" in what or "wgrad" in what:
    tf = tw = td = 0.0
    for name, lvl, k0, k1, N in FWD:
        M = LEVELS[lvl]; K = k0 + k1
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        dz = torch.randn(M, N, device=dev); st = ops.stat_buffer(M, N, K, dev)
        line = f"{name:9s} M={M:6d} K={K:4d} N={N:4d} "
        if "gemm" in what:
            t1 = timeit(lambda: ops.gemm(x, w, M, N, K, bias=b, stats=st))
            t2 = timeit(lambda: ops.linear_dgrad(dz, w))
            accbuf = torch.zeros(M, K, device=dev)
            t2a = timeit(lambda: ops.linear_dgrad(dz, w, acc=accbuf))  # input gradient ADDED to a shared buffer (GradSlot)
            byt = 4 * M * (K + N); fl = 2 * M * K * N
            line += f"fwd+stats {t1:7.1f}us ({byt/t1/1e3:6.0f} GB/s {fl/t1/1e6:6.1f} TF)  dgrad {t2:7.1f}us ({byt/t2/1e3:6.0f} GB/s)  dgrad+= {t2a:7.1f}us  "
            tf += t1; td += t2
        if "wgrad" in what:
            t3 = timeit(lambda: ops.linear_wgrad(dz, x, K))
            byt = 4 * M * (K + N); fl = 2 * M * K * N
            line += f"wgrad {t3:7.1f}us ({byt/t3/1e3:6.0f} GB/s {fl/t3/1e6:6.1f} TF)"
            tw += t3
        print(line)
    print(f"TOTAL fwd {tf:.0f} us, dgrad {td:.0f} us, wgrad {tw:.0f} us")
if "bnbwd" in what:
    # backward of every SharedMLP layer's BatchNorm + Linear input gradient: the column-sum pass (m3d_bn_bwd, reduce only)
    # and the fused dz-on-load input-gradient GEMM (m3d_bn_dgrad_f32), beside the plain input-gradient GEMM of the same shape
    tr = tf_ = tp = 0.0
    for name, lvl, k0, k1, N in FWD:
        if name in ("fc0", "fc_cls") or not ops.bn_dgrad_ok(N):
            continue
        M = LEVELS[lvl]; K = k0 + k1
        w = torch.randn(N, K, device=dev); dy = torch.randn(M, N, device=dev); z = torch.randn(M, N, device=dev)
        sc, sh, mu, isd = (torch.rand(N, device=dev) + 0.5 for _ in range(4))
        ns = ops.bn_bwd_slots(M)
        sums = torch.zeros((ns, 3, N), dtype=torch.float64, device=dev)
        def red():
            ops.call("m3d_bn_bwd", dy.data_ptr(), z.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), None, None,
                     None, None, None, 1, 0.2, M, N, sums.data_ptr(), None, None, None, None, None, None, 2 | (ns << 8),
                     None, torch.cuda.current_stream().cuda_stream)
        dx = torch.empty(M, K, device=dev); dz = torch.empty(M, N, device=dev); dg = torch.empty(N, device=dev); db = torch.empty(N, device=dev)
        def fused():
            ops.call("m3d_bn_dgrad_f32", dy.data_ptr(), z.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), 1, 0.2,
                     sums.data_ptr(), ns, M, N, w.data_ptr(), w.stride(0), K, dx.data_ptr(), K, dz.data_ptr(), dg.data_ptr(), db.data_ptr(),
                     0, 0, None, 0, None, torch.cuda.current_stream().cuda_stream)
        t1, t2, t3 = timeit(red), timeit(fused), timeit(lambda: ops.linear_dgrad(dz, w))
        byt_r = 8 * M * N; byt_f = 4 * M * (3 * N + K)
        print(f"{name:9s} M={M:6d} N={N:4d} Kin={K:4d}  reduce {t1:6.1f}us ({byt_r/t1/1e3:5.0f} GB/s)  fused dz+dgrad {t2:6.1f}us "
              f"({byt_f/t2/1e3:5.0f} GB/s)  plain dgrad {t3:6.1f}us")
        tr += t1; tf_ += t2; tp += t3
    print(f"TOTAL reduce {tr:.0f} us, fused {tf_:.0f} us, plain dgrad {tp:.0f} us")
if "bn" in what:
    for M, N in [(204800, 32), (204800, 64), (51200, 128), (12800, 256), (3200, 512)]:
        z = torch.randn(M, N, device=dev); dy = torch.randn(M, N, device=dev)
        sc, sh, mu, isd = (torch.rand(N, device=dev) + 0.5 for _ in range(4))
        t1 = timeit(lambda: ops.bn_apply(z, sc, sh, True))
        t2 = timeit(lambda: ops.bn_bwd(dy, z, sc, sh, mu, isd, True))
        print(f"bn M={M} N={N}: apply {t1:.1f}us ({8*M*N/t1/1e3:.0f} GB/s)  bwd {t2:.1f}us ({(4*3+4*2)*M*N/t2/1e3:.0f} GB/s alg 5 passes)")
if "knn" in what or "lfa" in what:
    from myria3d_amd.synthetic import synthetic_batch
    x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
    pos = pos.to(dev); ptr = ptr.to(dev)
    pos4 = ops.pad_pos(pos)
    plan_ptrs = [ptr]
    p4 = [pos4]
    for l in range(4):
        n = p4[-1].shape[0] // 4
        per = plan_ptrs[-1][1].item() // 4
        idx = (torch.arange(16, device=dev)[:, None] * (per * 4) + torch.stack([torch.randperm(per * 4, device=dev)[:per] for _ in range(16)])).reshape(-1).to(torch.int32)
        p4.append(ops.gather_rows(p4[-1], idx)); plan_ptrs.append(plan_ptrs[-1] // 4)
    idxs = []
    for l in range(4):
        tb = timeit(lambda: ops.KnnIndex(p4[l], plan_ptrs[l]))
        index = ops.KnnIndex(p4[l], plan_ptrs[l])
        tq = timeit(lambda: index.query(16, qry=index))
        idxs.append(index.query(16, qry=index)[0])
        n = p4[l].shape[0]
        print(f"knn level {l+1}: n={n} build {tb:.1f}us query(k=16) {tq:.1f}us  ({n*(16+64)/tq/1e3:.1f} GB/s algorithmic)")
        if l < 4:
            src = ops.KnnIndex(p4[l + 1], plan_ptrs[l + 1])
            t1 = timeit(lambda: src.query(1, qry=index))
            print(f"   1-NN into level {l+2}: {t1:.1f}us")
    if "lfa" in what:
        for l, (d_out) in enumerate([32, 128, 256, 512]):
            n = p4[l].shape[0]
            for ch in (d_out // 4, d_out // 2):
                D = ch // 2
                xin = torch.randn(n, D, device=dev); wf = torch.randn(D, 10, device=dev) * 0.3; bf = torch.randn(D, device=dev) * 0.1
                watt = torch.randn(ch, ch, device=dev) / ch ** 0.5
                tfw = timeit(lambda: ops.lfa_forward(xin, p4[l], idxs[l], wf, bf, watt, full=FULL))
                dout = torch.randn(n, ch, device=dev)
                def bwd():
                    dx = torch.zeros((n, D), device=dev); G = torch.empty(11 * D, dtype=torch.float64, device=dev)
                    dw = torch.empty((ch, ch), device=dev)
                    ws = torch.empty(ops.lib().m3d_lfa_bwd_workspace_bytes(n, 16, ch), dtype=torch.uint8, device=dev)
                    wp, wpt = ops.pack_attention_weight(watt), ops.pack_attention_weight(watt.t())
                    ops.call("m3d_lfa_bwd", xin.data_ptr(), p4[l].data_ptr(), idxs[l].data_ptr(), n, 16, ch, wf.data_ptr(), bf.data_ptr(),
                             wp.data_ptr(), wpt.data_ptr(), 0.2, dout.data_ptr(), dx.data_ptr(), dw.data_ptr(), 8 if FULL else 0, G.data_ptr(), ws.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
                tbw = timeit(bwd)
                alg = n * (16 + 4 * D + 64 + 4 * ch)
                fl = 2 * n * 16 * (ch * ch + 10 * D)
                print(f"lfa level {l+1} ch={ch:3d} n={n:6d}: fwd {tfw:7.1f}us ({alg/tfw/1e3:6.0f} GB/s, {fl/tfw/1e6:5.1f} TF)  bwd {tbw:7.1f}us ({3*fl/tbw/1e6:5.1f} TF)")
