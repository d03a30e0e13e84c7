"""Rates of the accumulate-update shapes of the factorizations (dst -= X Y^T, K = 512 / 1024, full and lower) next to
the dense product, whole chip, fp64.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
tag = " ".join(f"{k[9:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("FAER_HIP_") and k != "FAER_HIP_LIB")


def cm(m, n):
    return torch.randn((n, m), dtype=torch.float64, device="cuda").t()


def bench(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


n = 8192
a, b, c = cm(n, n), cm(n, n), cm(n, n)
ms = bench(lambda: F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0))
print(f"[{tag}] dgemm {n}^3 replace: {ms:.3f} ms {2 * n ** 3 / ms / 1e9:.1f} TF", flush=True)
del a, b, c
for r in (15360, 8192, 4096):
    for k in (512, 1024):
        x = cm(r, k)
        y = cm(r, k)
        c = cm(r, r)
        ms = bench(lambda: F.gemm(c, F.DST_FULL, F.ACCUM_ADD, x, y.t(), -1.0))
        print(f"[{tag}] full  r={r} k={k}: {ms:.3f} ms {2 * r * r * k / ms / 1e9:.1f} TF", flush=True)
        u = cm(k, r)  # the LU's trailing update: column-major U12 (K-major rhs)
        ms = bench(lambda: F.gemm(c, F.DST_FULL, F.ACCUM_ADD, x, u, -1.0))
        print(f"[{tag}] lu    r={r} k={k}: {ms:.3f} ms {2 * r * r * k / ms / 1e9:.1f} TF", flush=True)
        del u
        ms = bench(lambda: F.gemm(c, F.DST_LOWER, F.ACCUM_ADD, x, x.t(), -1.0))
        nt = r // 128
        fl = nt * (nt + 1) / 2 * 128 * 128 * k * 2
        print(f"[{tag}] lower r={r} k={k}: {ms:.3f} ms {fl / ms / 1e9:.1f} TF (tile flops)", flush=True)
        del x, y, c
