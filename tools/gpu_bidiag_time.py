"""timing of faer_hip_bidiag_in_place (device operands); algorithmic bytes: A22 read + written once and read once more
per column: sum_k 3 (m-k-1)(n-k-1) sizeof(T)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from gpu_util import init_gpu, to_dev
import torch

F = init_gpu()
for dtype in (np.float64, np.float32):
    for n in (1024, 4096):
        rng = np.random.default_rng(1)
        a = np.asarray(rng.standard_normal((n, n)), dtype=dtype, order="F")
        best = 1e9
        for rep in range(3):
            vd, hl, hr = to_dev(a), to_dev(np.zeros((32, n), dtype=dtype, order="F")), to_dev(np.zeros((32, n - 1), dtype=dtype, order="F"))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            F.bidiag_in_place(vd, hl, hr)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        k = np.arange(n, dtype=np.float64)
        byts = (3 * (n - k - 1) * (n - k - 1)).sum() * np.dtype(dtype).itemsize
        print(f"bidiag {np.dtype(dtype).name} n={n}: {best*1e3:.2f} ms, {byts/best/1e9:.0f} GB/s algorithmic, {8*n**3/3/best/1e9:.0f} GFLOP/s")
