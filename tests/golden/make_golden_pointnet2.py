"""Generates tests/golden/pointnet2_small.npz from oracle/pointnet2_oracle.py (run in the build container):

    python tests/golden/make_golden_pointnet2.py

The PointNet++ variant has NO reference implementation (myria3d/models/model.py:12), so these vectors pin the restated
oracle against itself over time (a regression fixture) and travel to the GPU box, where the HIP net is compared with them
directly; they do not pin anything to the reference.  Inputs follow the reference test's distribution (x, pos ~ U(0,1);
tests/myria3d/models/modules/test_randla_nets.py:25-26); parameters come from numpy's legacy RandomState.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pointnet2_oracle import PointNet2Oracle  # noqa: E402
from tests._util import fill_params_deterministic, rand_batch  # noqa: E402

SIZES = [300, 211, 12]  # the 12-point cloud has fewer points than K
K = 16
PARAM_SEED = 17


def main():
    torch.set_num_threads(1)
    x, pos, batch, ptr = rand_batch(SIZES, seed=77)
    n = sum(SIZES)
    mask = (np.random.RandomState(3).uniform(size=(n, 32)) > 0.5).astype(np.float32)
    y = torch.from_numpy(np.random.RandomState(4).randint(0, 6, (n,)))
    net = PointNet2Oracle(9, 6, num_neighbors=K, return_logits=True)
    fill_params_deterministic(net, PARAM_SEED)
    net.eval()
    with torch.no_grad():
        logits_eval = net(x, pos, batch, ptr)
    sel = [t.numpy().astype(np.int64) for t in net.last_sample_idx]
    net.train()
    logits_train = net(x, pos, batch, ptr, dropout_mask=torch.from_numpy(mask))
    loss = torch.nn.functional.cross_entropy(logits_train, y)
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters()}
    out = dict(x=x.numpy(), pos=pos.numpy(), ptr=ptr.numpy(), param_seed=np.int64(PARAM_SEED), k=np.int64(K),
               dropout_mask=mask, y=y.numpy(), logits_eval=logits_eval.numpy(), logits_train=logits_train.detach().numpy(),
               loss_train=np.float64(loss.item()),
               grad_sa1_lin0=grads["sa1.nn.lins.0.weight"].numpy(), grad_sa3_lin2=grads["sa3.nn.lins.2.weight"].numpy(),
               grad_fp1_lin0=grads["fp1.nn.lins.0.weight"].numpy(), grad_fc_classif=grads["fc_classif.weight"].numpy(),
               running_var_sa2_bn1=net.sa2.nn.norms[1].module.running_var.numpy())
    for i, s in enumerate(sel):
        out[f"fps{i}"] = s
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pointnet2_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
