"""GPU box: ``predict_cloud`` on bench.py's 10 M-point cloud with and without the one-batch lookahead of the position-only work
(round 6), interleaved repetitions on one box, the two results compared bit by bit; then the host's share of one call
(cProfile, top entries by cumulative time).  usage: python tools/predict_ab.py [reps] [batch_size ...]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myria3d_amd import predict_cloud

dev = torch.device("cuda:0")
# stream -> hardware-queue aliasing probe: torch hands out its pool streams in turn, ROCclr maps them onto 4 hardware queues
_dummies = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("M3D_DUMMY_STREAMS", "0")))]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
net, pos, x = bench.predict_e2e_inputs(dev)


sizes = [int(a) for a in sys.argv[2:]] or [50]
bs = sizes[0]


def run(look, seed=1234):
    net.set_decimation_seed(seed)
    return predict_cloud(net, pos, x, tile_width=1000.0, subtile_width=50, batch_size=bs, lookahead=look)


for bs in sizes:
    outs = {look: run(look) for look in (False, True)}
    a, b = outs[False], outs[True]
    print(f"batch_size {bs}: bitwise equal logits:", bool(torch.equal(a["logits_full"], b["logits_full"])), "preds:", bool(torch.equal(a["preds"], b["preds"])))
    del outs, a, b
    T = {False: [], True: []}
    for r in range(reps):
        for look in (False, True):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run(look)
            torch.cuda.synchronize(); T[look].append((time.perf_counter() - t0) * 1e3)
    for look in (False, True):
        v = sorted(T[look])
        print(f"batch_size {bs} lookahead={look}: min {v[0]:.2f} median {v[len(v) // 2]:.2f} ms per 10 M points   {['%.2f' % t for t in T[look]]}")
if os.environ.get("M3D_PREDICT_HOST_PROFILE"):
    pr = cProfile.Profile()
    pr.enable(); run(True); torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
