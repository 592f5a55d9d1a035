#!/bin/bash
# GPU box: PointNet++ variant parity + FPS timings
set -u
timeout -s KILL 500 python -m pytest tests/test_gpu_pointnet2.py -m gpu -q --durations=3 2>&1 | tail -12
timeout -s KILL 200 python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from myria3d_amd import ops
from myria3d_amd.synthetic import synthetic_batch
dev = torch.device('cuda:0')
for n, B in ((12800, 16), (40000, 16), (40000, 1), (10000, 16), (3200, 16)):
    x, pos, batch, ptr, y = synthetic_batch([n] * B)
    pos4 = ops.pad_pos(pos.to(dev)); ptr = ptr.to(dev); out = (ptr // 4).contiguous()
    for _ in range(2): ops.fps(pos4, ptr, out, int(out[-1]), n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.fps(pos4, ptr, out, int(out[-1]), n); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    print(f"fps {B} x {n} -> {n // 4}: {ms:.2f} ms  ({ms * 1e3 / (n // 4):.2f} us per iteration)")
PY
