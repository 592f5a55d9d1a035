"""Generates tests/golden/randla_small.npz from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

The reference itself cannot be imported here (torch_geometric / torch_cluster / torch_scatter are not installed,
SURVEY.md §0), so these vectors pin the *oracle* (oracle/randla_oracle.py) — parity with the reference's own
implementation stays unpinned, see oracle/__init__.py.  Inputs follow the reference test's distribution
(x, pos ~ U(0,1); tests/myria3d/models/modules/test_randla_nets.py:25-26); parameters come from numpy's legacy
RandomState via tests/_util.fill_params_deterministic so nothing depends on torch's RNG stream.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.randla_oracle import RandLANetOracle, fixed_decimation_indices  # noqa: E402
from tests._util import fill_params_deterministic, rand_batch  # noqa: E402

SIZES = [300, 211]
PARAM_SEED = 42


def main():
    torch.set_num_threads(1)
    x, pos, batch, ptr = rand_batch(SIZES, seed=2024)
    dec = fixed_decimation_indices(ptr.tolist(), 4, seed=5)
    n = sum(SIZES)
    mask = (np.random.RandomState(9).uniform(size=(n, 32)) > 0.5).astype(np.float32)
    net = RandLANetOracle(9, 6, return_logits=True)
    fill_params_deterministic(net, PARAM_SEED)
    net.eval()
    rec = {}
    with torch.no_grad():
        logits_eval = net(x, pos, batch, ptr, decimation_idx=dec, record=rec)
    net.train()
    y = torch.from_numpy(np.random.RandomState(10).randint(0, 6, (n,)))
    logits_train = net(x, pos, batch, ptr, decimation_idx=dec, dropout_mask=torch.from_numpy(mask))
    loss = torch.nn.functional.cross_entropy(logits_train, y)
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters()}
    out = dict(
        x=x.numpy(), pos=pos.numpy(), ptr=ptr.numpy(), param_seed=np.int64(PARAM_SEED), dropout_mask=mask,
        y=y.numpy(), logits_eval=logits_eval.numpy(), logits_train=logits_train.detach().numpy(),
        loss_train=np.float64(loss.item()), knn_idx_level1=rec["block1.knn_idx"].numpy().astype(np.int32),
        grad_fc0_weight=grads["fc0.weight"].numpy(), grad_fc_classif_weight=grads["fc_classif.weight"].numpy(),
        grad_block1_lfa1_att=grads["block1.lfa1.mlp_attention.lins.0.weight"].numpy(),
        grad_block4_lfa2_enc=grads["block4.lfa2.mlp_encoder.lins.0.weight"].numpy(),
        running_mean_block1_lfa1_enc=net.block1.lfa1.mlp_encoder.norms[0].module.running_mean.numpy(),
    )
    for i, d in enumerate(dec):
        out[f"dec{i}"] = d.numpy().astype(np.int64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "randla_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
