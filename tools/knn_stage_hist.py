"""Evidence for the staged kNN (csrc/knn.hip: knn_stage_kernel) at BASELINE config 2, level 1 (16 x 12 800 points, K = 16):
how many queries are still OPEN after ring r (the pool counter of a two-stage run "r,16"), i.e. the distribution of the
ring at which a query's search closes — the histogram the tail argument rests on.  Run it under
``rocprofv3 --kernel-trace --stats`` to get the duration of every stage kernel beside it (tools/gpu_r03_c.sh)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import ops
from myria3d_amd._lib import call, lib
from myria3d_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
ix = ops.KnnIndex(pos.to(dev), ptr.to(dev))
n, K = ix.n, 16
scratch = torch.empty(lib().m3d_knn_staged_workspace_bytes(n, K), dtype=torch.uint8, device=dev)
idx = torch.empty((n, K), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(schedule):
    os.environ["M3D_KNN_STAGES"] = schedule
    call("m3d_knn_query_staged", ix.ws.data_ptr(), ix.ptr.data_ptr(), n, ix.num_clouds, ix.ws.data_ptr(), ix.ptr.data_ptr(),
         n, K, 1, idx.data_ptr(), None, scratch.data_ptr(), st)
    torch.cuda.synchronize()
    return scratch[:16].view(torch.int32).tolist()


prev = n
print(f"level 1: {n} queries, K = {K}; open = search not closed after ring r (exact termination test)")
for r in range(0, 9):
    c = run(f"{r},16")[0]
    print(f"  after ring {r}: open {c:7d} = {100.0 * c / n:6.2f} %   closed at this ring {prev - c:7d} = {100.0 * (prev - c) / n:6.2f} %")
    prev = c
    if c == 0:
        break
for sched in ("2,4:3,8:4,16", "2,4:3,16", "2,16"):
    print(sched, "-> survivors per stage", run(sched)[:3])
