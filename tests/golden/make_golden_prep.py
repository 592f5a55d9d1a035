"""Generates tests/golden/prep_small.npz from the CPU oracles (run in the build container):

    python tests/golden/make_golden_prep.py

Golden vectors for the two "next" boundaries around the network:
  * data preparation: GridSampling(0.25) -> Center -> NullifyLowestZ -> NormalizePos -> StandardizeRGBAndIntensity
    of two small raw tiles (oracle/prep_oracle.py);
  * merged predictions: Interpolator arithmetic on three overlapping tiles (oracle.randla_oracle.interpolator_reduce).
Like randla_small.npz they pin the ORACLES (GridSampling / Center are restated PyG behaviour: parity with the reference
stays unpinned); inputs come from numpy's legacy RandomState only.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import prep_oracle as P  # noqa: E402
from oracle.randla_oracle import interpolator_reduce  # noqa: E402


def inputs():
    rs = np.random.RandomState(77)
    sizes = [900, 400]
    pos = np.concatenate([rs.uniform(0, 1, (n, 3)) * [6.0, 5.0, 2.0] + [100.0 * i, -40.0 * i, 10.0 * i]
                          for i, n in enumerate(sizes)]).astype(np.float32)
    x = rs.uniform(0, 1, (sum(sizes), 9)).astype(np.float32)
    x[:, 0] = rs.gamma(2.0, 300.0, sum(sizes)).astype(np.float32)
    x[:, 7] = rs.uniform(0, 255, sum(sizes)).astype(np.float32)
    y = rs.randint(0, 6, sum(sizes)).astype(np.int64)
    ptr = np.array([0, 900, 1300], np.int64)
    nb_points, C = 700, 7
    logits = [rs.normal(0, 3, (m, C)).astype(np.float32) for m in (300, 250, 200)]
    idx = [rs.choice(nb_points, m, replace=False).astype(np.int64) for m in (300, 250, 200)]
    return pos, x, y, ptr, logits, idx, nb_points


def main():
    torch.set_num_threads(1)
    pos, x, y, ptr, logits, idx, nb_points = inputs()
    p, xx, yy, optr = P.prepare_tiles(torch.from_numpy(pos), torch.from_numpy(x), torch.from_numpy(y), ptr.tolist(),
                                      0.25, 50, 0, 7)
    rows, probas, preds, entropy, cat_idx = interpolator_reduce([torch.from_numpy(l) for l in logits], idx, nb_points)
    out = dict(pos=pos, x=x, y=y, ptr=ptr, prep_pos=p.numpy(), prep_x=xx.numpy(), prep_y=yy.numpy(),
               prep_ptr=np.asarray(optr, np.int64), nb_points=np.int64(nb_points), rows=rows.numpy(),
               probas=probas.numpy(), preds=preds.numpy(), entropy=entropy.numpy(), cat_idx=cat_idx.numpy())
    for i, (l, j) in enumerate(zip(logits, idx)):
        out[f"logits{i}"], out[f"idx{i}"] = l, j
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prep_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
