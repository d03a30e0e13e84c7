"""helpers shared by the -m gpu parity tests (device tensors are column major like faer::Mat)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402

EPS = {np.dtype(np.float64): np.finfo(np.float64).eps, np.dtype(np.float32): np.finfo(np.float32).eps}


def fa():
    return ge.load_package()


def init_gpu():
    import torch

    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    m = fa()
    m.lib()
    m.use_torch_stream()
    return m


def to_dev(x, order="F"):
    """numpy 2-D array -> torch cuda tensor with the requested memory order (same logical values)."""
    import torch

    x = np.asarray(x)
    if order == "F":
        return torch.from_numpy(np.ascontiguousarray(x.T)).cuda().t()
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def to_host(t):
    fa().synchronize()
    return t.detach().cpu().numpy()


def rnd(rng, m, n, dtype=np.float64, order="F"):
    return np.asarray(rng.standard_normal((m, n)), dtype=dtype, order=order)


def spd(rng, n, dtype=np.float64):
    a = rng.standard_normal((n, n))
    return np.asarray(a @ a.T + n * np.eye(n), dtype=dtype, order="F")
