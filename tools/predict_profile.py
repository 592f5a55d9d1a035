"""GPU box: where the time of myria3d_amd.predict_cloud goes — every stage of the chain timed with a device synchronize
behind it (so the figures are GPU time + launch overhead of that stage alone), on bench.py's 10 M-point cloud.
usage: python tools/predict_profile.py [points_per_m2]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from myria3d_amd import HipRandLANet, ops
from myria3d_amd.interpolation import DeviceInterpolator, knn_interpolate
from myria3d_amd.predict import itp_reduce
from myria3d_amd.tiling import tile_select
from myria3d_amd.transforms import grid_sampling, node_budget, normalize_tiles

dev = torch.device("cuda:0")
side, density = 1000.0, float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
rs = np.random.RandomState(0)
n = int(side * side * density)
xy = rs.uniform(0, side, (n, 2)).astype(np.float32)
z = (2.0 * np.sin(2 * np.pi * xy[:, 0] / 50.0) + 3 * rs.uniform(size=n)).astype(np.float32)
pos = torch.from_numpy(np.concatenate([xy, z[:, None]], 1)).to(dev)
x = torch.rand((n, 9), device=dev)
torch.manual_seed(0)
net = HipRandLANet(9, 7, num_neighbors=16, return_logits=True).to(dev).eval()
T = {}
def tick(name, t0):
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return time.perf_counter()
with torch.no_grad():
    for rep in range(2):
        T.clear()
        torch.cuda.synchronize(); t = time.perf_counter()
        sample_ptr, idx, _ = tile_select(pos, side, 50, 0); t = tick("tile_select", t)
        bounds = sample_ptr.tolist(); t = tick("sample_ptr.tolist", t)
        samples = [s for s in range(len(bounds) - 1) if bounds[s + 1] > bounds[s]]
        itp = DeviceInterpolator()
        for b0 in range(0, len(samples), 50):
            chunk = samples[b0:b0 + 50]
            rows = torch.cat([idx[bounds[s]:bounds[s + 1]] for s in chunk])
            sizes = torch.tensor([bounds[s + 1] - bounds[s] for s in chunk], dtype=torch.int64)
            ptr_full = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)]).to(dev); t = tick("rows cat + ptr", t)
            pos_copy = ops.gather_rows(pos, rows); x_raw = ops.gather_rows(x, rows); t = tick("gather rows", t)
            p, xx, _, ptr = grid_sampling(pos_copy, x_raw, None, ptr_full, 0.25); t = tick("grid_sampling", t)
            p, xx, _, ptr, _ = node_budget(p, xx, None, ptr, minimum=300, maximum=40000, seed=b0); t = tick("node_budget", t)
            pn, xn = normalize_tiles(p, xx, ptr, center=True, nullify_z=True, subtile_width=50, intensity_col=0, rgb_col=7); t = tick("normalize", t)
            logits = net(xn, pn, None, ptr); t = tick("net forward", t)
            cnt = ptr[1:] - ptr[:-1]
            bx = torch.repeat_interleave(torch.arange(len(chunk), device=dev), cnt)
            by = torch.repeat_interleave(torch.arange(len(chunk), device=dev), sizes.to(dev)); t = tick("batch vectors", t)
            full = knn_interpolate(logits, p, pos_copy, batch_x=bx, batch_y=by, k=10); t = tick("knn_interpolate", t)
            itp.store_predictions(full, rows)
        out = itp_reduce(itp, n); t = tick("merge + softmax", t)
    tot = sum(T.values())
    for k, v in T.items():
        print(f"{k:22s} {v:8.2f} ms")
    print(f"{'sum (synchronised)':22s} {tot:8.2f} ms for {n} points, {len(samples)} samples")
