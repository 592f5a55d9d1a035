"""per-wavefront work of the cooperative self-kNN at the BASELINE config-2 level shapes; needs the instrumented variant:
tools/build_variant.sh knnc_stats knn.hip -DKNNC_STATS && M3D_LIB=$PWD/myria3d_amd/variants/libm3d_knnc_stats.so python tools/knn_stats.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myria3d_amd import ops
from myria3d_amd._lib import lib
from myria3d_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
pos, ptr = pos.to(dev), ptr.to(dev)
p4, ptrs = [ops.pad_pos(pos)], [ptr]
g = torch.Generator(device=dev).manual_seed(0)
for l in range(3):
    per = int(ptrs[-1][1].item())
    idx = torch.cat([b * per + torch.randperm(per, device=dev, generator=g)[: per // 4] for b in range(16)]).to(torch.int32)
    p4.append(ops.gather_rows(p4[-1], idx)); ptrs.append(ptrs[-1] // 4)
h = lib()
h.m3d_knn_debug_stats.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 8)()
for l in range(4):
    index = ops.KnnIndex(p4[l], ptrs[l])
    torch.cuda.synchronize()
    h.m3d_knn_debug_stats(buf, 1)
    index.query(16, qry=index, sorted_io=True)
    torch.cuda.synchronize()
    h.m3d_knn_debug_stats(buf, 1)
    w = max(buf[0], 1)
    print(f"knn_stats L{l+1} target={os.environ.get('M3D_KNN_CELL_TARGET','7')}: waves={buf[0]} segments/wave={buf[1]/w:.2f} rings/wave={buf[2]/w:.2f} "
          f"chunks/wave={buf[3]/w:.1f} candidates/wave={buf[4]/w:.0f} chain_trips/wave={buf[5]/w:.0f} appends/lane={buf[6]/w/64:.1f}")
