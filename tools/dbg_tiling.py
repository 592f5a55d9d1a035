import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
from oracle import prep_oracle as O
from myria3d_amd import tiling
spec = importlib.util.spec_from_file_location('tt', 'tests/test_tiling.py'); tt = importlib.util.module_from_spec(spec); spec.loader.exec_module(tt)
n, tile, sub, ov = 200000, 1000, 50, 0
pos = tt._cloud(n, tile, seed=n + ov, lattice=False)
ref = {s: np.sort(i) for s, i in O.split_cloud_into_samples(pos, tile, sub, ov)}
sp, ix, c = tiling.tile_select(torch.from_numpy(pos).cuda(), tile, sub, ov)
sp, ix = sp.cpu().numpy(), ix.cpu().numpy()
print("total", sp[-1], sum(len(v) for v in ref.values()))
bad = 0
for s in range(len(sp) - 1):
    got = ix[sp[s]:sp[s + 1]]; want = ref.get(s, np.zeros(0, int))
    if len(got) != len(want) or not np.array_equal(got, want):
        bad += 1
        if bad <= 5:
            print("sample", s, "len", len(got), len(want), "missing", sorted(set(want) - set(got))[:8], "extra", sorted(set(got) - set(want))[:8],
                  "sorted?", bool((np.diff(got) > 0).all()))
print("bad samples", bad)
