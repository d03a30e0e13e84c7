"""one tridiagonalization per size for a kernel trace: rocprofv3 --kernel-trace --stats -- python tools/gpu_tridiag_prof.py [n ...]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gpu_util import init_gpu, to_dev
import torch

F = init_gpu()
for n in [int(v) for v in sys.argv[1:]] or [4096]:
    rng = np.random.default_rng(1)
    a = rng.standard_normal((n, n)); a = np.asarray(a + a.T, dtype=np.float64, order="F")
    for rep in range(2):
        vd, hd = to_dev(a), to_dev(np.zeros((32, n - 1), dtype=np.float64, order="F"))
        torch.cuda.synchronize()
        F.tridiag_in_place(vd, hd)
        torch.cuda.synchronize()
