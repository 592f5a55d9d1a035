"""Kernel-by-kernel timeline of ONE graphed eval forward (BASELINE config 2): run under
``rocprofv3 --kernel-trace --output-format csv -d DIR -o trace -- python tools/eval_trace.py [bf16]``, then
``python tools/eval_trace.py --analyze DIR/…kernel_trace.csv OUT.csv`` (a torch fill between steps marks the boundaries)."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "--analyze":
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""),
                 r.get("Grid_Size_X", r.get("Grid_Size", ""))) for r in rows)
    marks = [i for i, e in enumerate(ev) if "FillFunctor<long>" in e[2] and e[4] in ("1", "64", "128", "256")]
    a, b = marks[-3], marks[-2]
    seg = ev[a + 1:b]
    t0 = seg[0][0]
    with open(sys.argv[3], "w") as fh:
        fh.write("start_us,dur_us,queue,grid_x,kernel\n")
        for s_, e_, n_, q_, g_ in seg:
            fh.write(f"{(s_ - t0) / 1e3:.1f},{(e_ - s_) / 1e3:.1f},{q_},{g_},{n_.split('(')[0].replace('void ', '').replace(',', ';').replace(' ', '')[:100]}\n")
    per = {}
    for s_, e_, n_, q_, g_ in seg:
        per.setdefault(q_, [0.0, s_, e_])
        per[q_][0] += (e_ - s_) / 1e3
        per[q_][2] = e_
    print(f"eval forward: {len(seg)} kernels, wall {(seg[-1][1] - t0) / 1e3:.1f} us; per queue (busy us, first start, last end): "
          + str({q: (round(v[0], 1), round((v[1] - t0) / 1e3, 1), round((v[2] - t0) / 1e3, 1)) for q, v in per.items()}))
    sys.exit(0)

import torch

from myria3d_amd import GraphedStep, HipRandLANet
from myria3d_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
x, pos, batch, ptr, y = synthetic_batch([12800] * 16)
torch.manual_seed(0)
net = HipRandLANet(9, 6, return_logits=True).to(dev).eval()
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    net.matmul_precision, net.activation_dtype = "bf16", torch.bfloat16
gs = GraphedStep(net, ptr, 9, mode="eval")
gs.load_all(x.to(dev), pos.to(dev))
gs.prepare()
mark = torch.zeros(1, dtype=torch.int64, device=dev)
for _ in range(8):
    gs.step()
    mark.fill_(0)
torch.cuda.synchronize()
