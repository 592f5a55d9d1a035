"""Shared test helpers (deterministic parameters independent of torch's RNG stream)."""
import numpy as np
import torch


def fill_params_deterministic(module: torch.nn.Module, seed: int = 0) -> None:
    """Fill every parameter/buffer from numpy's legacy RandomState (stable across versions/platforms):
    Linear weights/biases ~ U(+-1/sqrt(fan_in)); BN gamma ~ U(.5,1.5), beta ~ U(-.2,.2), running_mean ~ U(-.2,.2),
    running_var ~ U(.5,1.5)."""
    rs = np.random.RandomState(seed)
    sd = module.state_dict()
    new = {}
    for name in sorted(sd.keys()):
        t = sd[name]
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            new[name] = torch.zeros_like(t)
        elif name.endswith("running_var"):
            new[name] = torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
        elif name.endswith("running_mean"):
            new[name] = torch.from_numpy(rs.uniform(-0.2, 0.2, shape).astype(np.float32))
        elif "norms." in name and name.endswith("weight"):
            new[name] = torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
        elif "norms." in name and name.endswith("bias"):
            new[name] = torch.from_numpy(rs.uniform(-0.2, 0.2, shape).astype(np.float32))
        elif name.endswith("weight"):
            b = 1.0 / np.sqrt(shape[1])
            new[name] = torch.from_numpy(rs.uniform(-b, b, shape).astype(np.float32))
        elif name.endswith("bias"):
            # fan_in of the matching weight
            w = sd[name[: -len("bias")] + "weight"]
            b = 1.0 / np.sqrt(w.shape[1])
            new[name] = torch.from_numpy(rs.uniform(-b, b, shape).astype(np.float32))
        else:
            new[name] = t
    module.load_state_dict(new)


def rand_batch(sizes, num_features=9, seed=0):
    """The reference test's input distribution: x, pos ~ U(0,1) (tests/myria3d/models/modules/test_randla_nets.py:25-26)."""
    rs = np.random.RandomState(seed)
    n = sum(sizes)
    x = torch.from_numpy(rs.uniform(0, 1, (n, num_features)).astype(np.float32))
    pos = torch.from_numpy(rs.uniform(0, 1, (n, 3)).astype(np.float32))
    ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    return x, pos, batch, ptr
