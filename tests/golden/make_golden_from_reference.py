"""Pins the oracle (and, on the GPU box, the HIP net) to the reference's OWN file: generates ``randla_reference.npz``.

    python tests/golden/make_golden_from_reference.py  [/path/to/myria3d checkout, default /root/reference]

The script imports ``PyGRandLANet`` from ``myria3d/models/modules/pyg_randla_net.py`` of the checkout (by file path: the
package ``__init__`` pulls in Lightning / hydra, which the net itself does not need).  The six third-party symbols that file
imports come from the real wheels (``torch_geometric`` 2.4, ``torch_cluster``, ``torch_scatter``; environment.yml:14-22 of the
reference) where they are installed, else from ``tests/_pyg_stub`` (a restatement of exactly those symbols, SURVEY Appendix
A); the ``stack`` entry of the file records which.  HAS BEEN RUN in the build container (rounds 3 and 4, stub stack: the wheels
cannot be installed there); the committed file is its output and ``tests/test_reference_pin.py`` regenerates and diffs it.

For every size set it
  1. loads the deterministic weights of ``tests/_util.fill_params_deterministic`` (state_dict keys are shared by construction,
     SURVEY 8b), injects fixed decimation indices by replacing the module-level ``decimation_indices``
     (pyg_randla_net.py:192-231) and switches the classifier dropout off (PyG ``MLP.dropout`` is a plain list; torch's
     dropout stream cannot be injected),
  2. writes the reference's outputs: eval logits, train-mode logits, loss, EVERY parameter gradient (141 tensors, round 4;
     seven in round 3), running statistics, the level-1 kNN edge list as per-centre squared distances,
  3. (round 6) runs the same module in fp64 and writes its train logits, loss and every gradient (``grad64:`` keys).
Size sets: ``[700, 333, 50]`` (keys without prefix) and, round 4, ``[300, 9, 1, 120]`` (prefix ``b/``): a cloud with fewer
points than K = 16 and a one-point cloud (CHANGELOG 3.4.0 of the reference: tiles down to one point).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests._util import fill_params_deterministic, rand_batch  # noqa: E402

SIZES = [700, 333, 50]
SIZE_SETS = {"": [700, 333, 50], "b/": [300, 9, 1, 120]}  # key prefix -> tile sizes
PARAM_SEED = 77
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "randla_reference.npz")
GRAD_KEYS = ["fc0.weight", "block1.lfa1.mlp_attention.lins.0.weight", "block1.lfa2.mlp_encoder.lins.0.weight",
             "block2.mlp2.norms.0.module.weight", "block4.lfa2.mlp_encoder.lins.0.weight", "fp2.nn.lins.0.weight",
             "fc_classif.weight"]


STUB_DIR = os.path.join(ROOT, "tests", "_pyg_stub")


def load_reference_module(ref_root: str, allow_stub: bool = True):
    """The reference's net module, imported by file path.  With the real PyG stack importable it runs on that
    (``mod.M3D_STACK == "real PyG wheels"``); otherwise — and with ``allow_stub`` — on ``tests/_pyg_stub`` (a
    restatement of the six third-party symbols the file needs; ``mod.M3D_STACK == "reference file + stub PyG"``).
    Raises ImportError when there is no checkout (the GPU box) or no usable stack."""
    path = os.path.join(ref_root, "myria3d", "models", "modules", "pyg_randla_net.py")
    if not os.path.exists(path):
        raise ImportError(f"no reference checkout at {ref_root}")

    def load():
        spec = importlib.util.spec_from_file_location("_m3d_reference_pyg_randla_net", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)  # ImportError here = torch_geometric / torch_cluster / torch_scatter missing
        return mod

    try:
        mod = load()
        import torch_geometric

        mod.M3D_STACK = "reference file + stub PyG" if getattr(torch_geometric, "IS_M3D_STUB", False) else "real PyG wheels"
    except ImportError:
        if not allow_stub:
            raise
        for name in [m for m in sys.modules if m.split(".")[0] in ("torch_geometric", "torch_scatter", "torchmetrics")]:
            del sys.modules[name]  # (a half-imported real stack must not shadow the stub)
        sys.path.insert(0, STUB_DIR)
        mod = load()
        mod.M3D_STACK = "reference file + stub PyG"
    return mod


def fixed_decimation(ptr, factor, levels, seed):
    """Same indices as oracle.randla_oracle.fixed_decimation_indices, without importing the oracle."""
    g = torch.Generator().manual_seed(seed)
    out, p = [], [int(v) for v in ptr]
    for _ in range(levels):
        idx, new = [], [0]
        for b in range(len(p) - 1):
            n = p[b + 1] - p[b]
            m = max(1, n // factor)
            idx.append(p[b] + torch.randperm(n, generator=g)[:m])
            new.append(new[-1] + m)
        out.append(torch.cat(idx))
        p = new
    return out


def run_reference(mod, x, pos, batch, ptr, dec, y, param_seed=PARAM_SEED, dtype=torch.float32):
    """Eval logits, train logits (dropout off), loss, gradients, running statistics of the REAL PyGRandLANet.
    ``dtype=torch.float64``: the same module in double precision (round 6: the yardstick the fp32 run AND the HIP net are
    measured against — how much of a gradient's difference is the fp32 arithmetic's own rounding)."""
    net = mod.PyGRandLANet(9, 6, decimation=4, num_neighbors=16, return_logits=True)
    fill_params_deterministic(net, param_seed)
    net.mlp_classif.dropout = [0.0, 0.0]
    net = net.to(dtype)
    x, pos = x.to(dtype), pos.to(dtype)
    calls = {"i": 0}
    orig = mod.decimation_indices

    def injected(ptr_in, factor):
        idx = dec[calls["i"] % len(dec)]
        calls["i"] += 1
        _, ptr_out = orig(ptr_in, factor)  # the reference's own ptr arithmetic (deterministic)
        assert idx.numel() == int(ptr_out[-1])
        return idx, ptr_out

    mod.decimation_indices = injected
    try:
        net.eval()
        with torch.no_grad():
            logits_eval = net(x, pos, batch, ptr)
        net.train()
        logits_train = net(x, pos, batch, ptr)
        loss = torch.nn.functional.cross_entropy(logits_train, y)
        loss.backward()
    finally:
        mod.decimation_indices = orig
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    bufs = {k: b.detach().clone() for k, b in net.named_buffers() if k.endswith(("running_mean", "running_var"))}
    # level-1 graph through the reference's own knn_graph (torch_cluster): per-centre ascending squared distances
    edge = mod.knn_graph(pos, 16, batch=batch, loop=True)
    d2 = ((pos[edge[0]] - pos[edge[1]]) ** 2).sum(1)
    return dict(logits_eval=logits_eval, logits_train=logits_train.detach(), loss=loss.detach(), grads=grads, bufs=bufs,
                knn_src=edge[0], knn_dst=edge[1], knn_d2=d2)


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("M3D_REFERENCE_ROOT", "/root/reference")
    mod = load_reference_module(ref_root)
    torch.set_num_threads(1)
    out = dict(param_seed=np.int64(PARAM_SEED), stack=np.array(mod.M3D_STACK), sets=np.array(list(SIZE_SETS)))
    for pre, sizes in SIZE_SETS.items():
        x, pos, batch, ptr = rand_batch(sizes, seed=2025 + len(pre))
        dec = fixed_decimation(ptr.tolist(), 4, 4, seed=8)
        y = torch.from_numpy(np.random.RandomState(3).randint(0, 6, (sum(sizes),)))
        r = run_reference(mod, x, pos, batch, ptr, dec, y)
        out.update({pre + "x": x.numpy(), pre + "pos": pos.numpy(), pre + "ptr": ptr.numpy(), pre + "y": y.numpy(),
                    pre + "logits_eval": r["logits_eval"].numpy(), pre + "logits_train": r["logits_train"].numpy(),
                    pre + "loss_train": np.float64(r["loss"].item()), pre + "knn_src": r["knn_src"].numpy(),
                    pre + "knn_dst": r["knn_dst"].numpy(), pre + "knn_d2": r["knn_d2"].numpy()})
        for i, d in enumerate(dec):
            out[f"{pre}dec{i}"] = d.numpy().astype(np.int64)
        for k, g in r["grads"].items():  # every parameter gradient
            out[pre + "grad:" + k] = g.numpy()
        # round 6: the reference module once more in fp64 — every gradient (stored rounded to fp32: 6e-8 relative) and the
        # train-mode logits / loss.  Tests bound the HIP net's error against THIS run by max(1e-3, 2 x the fp32 run's own error)
        r64 = run_reference(mod, x, pos, batch, ptr, dec, y, dtype=torch.float64)
        out[pre + "logits_train64"] = r64["logits_train"].float().numpy()
        out[pre + "loss_train64"] = np.float64(r64["loss"].item())
        for k, g in r64["grads"].items():
            out[pre + "grad64:" + k] = g.float().numpy()
        for k, b in r["bufs"].items():
            if k.startswith(("block1.lfa1.mlp_encoder", "block3.mlp2", "mlp_summit")):
                out[pre + "buf:" + k] = b.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes — generated from the reference at", ref_root, "on:", mod.M3D_STACK)


if __name__ == "__main__":
    main()
