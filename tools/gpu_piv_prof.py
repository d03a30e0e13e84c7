"""one full-pivot LU / column-pivot QR for a kernel trace: rocprofv3 --kernel-trace --stats -- python tools/gpu_piv_prof.py <fplu|cpqr> [n]"""
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
for rep in range(2):
    vd = to_dev(a)
    torch.cuda.synchronize()
    if name == "fplu":
        F.full_piv_lu_factor_in_place(vd)
    else:
        h = to_dev(np.zeros((F.qr_recommended_block_size(n, n, np.float64), n), dtype=np.float64, order="F"))
        F.colpiv_qr_factor_in_place(vd, h)
    torch.cuda.synchronize()
