"""one level-2 reduction per name for a kernel trace: rocprofv3 --kernel-trace --stats -- python tools/gpu_l2_prof.py <tridiag|bidiag|hess> [n]"""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gpu_util import init_gpu, to_dev
import torch

F = init_gpu()
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(1)
a = np.asarray(rng.standard_normal((n, n)), dtype=np.float64, order="F")
if name == "tridiag":
    a = np.asfortranarray(a + a.T)
for rep in range(2):
    vd = to_dev(a)
    h1, h2 = to_dev(np.zeros((32, n), dtype=np.float64, order="F")), to_dev(np.zeros((32, n - 1), dtype=np.float64, order="F"))
    torch.cuda.synchronize()
    if name == "tridiag":
        F.tridiag_in_place(vd, h2)
    elif name == "bidiag":
        F.bidiag_in_place(vd, h1, h2)
    else:
        F.hessenberg_in_place(vd, h2)
    torch.cuda.synchronize()
