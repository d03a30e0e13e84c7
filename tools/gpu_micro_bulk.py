"""Dense / SYRK-shaped GEMM rates on the whole chip vs on the CU-masked bulk stream of the look-ahead drivers."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
L.faer_hip_debug_internal_stream.restype = C.c_void_p


def cm(m, n):
    return torch.randn((n, m), dtype=torch.float64, device="cuda").t()


def bench(fn, reps=8):
    fn()
    F.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    F.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n = 8192
a, b, c = cm(n, n), cm(n, n), cm(n, n)
r, k = 15360, 1024
x, cc = cm(r, k), cm(r, r)
nt = r // 128
fl = nt * (nt + 1) / 2 * 128 * 128 * k * 2
torch.cuda.synchronize()
for name, which in (("default stream", 0), ("bulk stream (224 CUs)", 1), ("panel stream (32 CUs)", 2)):
    if which:
        L.faer_hip_set_stream(C.c_void_p(L.faer_hip_debug_internal_stream(which)))
    if which < 2:
        ms = bench(lambda: F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0))
        print(f"{name}: dgemm 8192^3 {ms:.3f} ms {2 * n ** 3 / ms / 1e9:.1f} TF", flush=True)
        ms = bench(lambda: F.gemm(cc, F.DST_LOWER, F.ACCUM_ADD, x, x.t(), -1.0))
        print(f"{name}: lower r={r} k={k}: {ms:.3f} ms {fl / ms / 1e9:.1f} TF", flush=True)
    m2 = 2048
    a2, b2, c2 = cm(m2, m2), cm(m2, m2), cm(m2, m2)
    ms = bench(lambda: F.matmul(c2, F.ACCUM_ADD, a2, b2, 1.0))
    print(f"{name}: dgemm 2048^3 add {ms:.3f} ms {2 * m2 ** 3 / ms / 1e9:.1f} TF", flush=True)
