"""Rate of the trailing-update shape of the blocked Cholesky (C(lower) -= X X^T, K = 1024) against the dense
kernel's rate on the same box: tile-order / quantisation effects in isolation.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()


def cm(m, n):
    return torch.randn((n, m), dtype=torch.float64, device="cuda").t()


def bench(fn, reps=5):
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
print(f"dgemm {n}^3 replace: {ms:.3f} ms {2 * n ** 3 / ms / 1e9:.1f} TF", flush=True)
del a, b, c
for r in (15360, 12288, 8192, 4096):
    for k in (1024, 512, 2048):
        x = cm(r, k)
        c = cm(r, r)
        ms = bench(lambda: F.gemm(c, F.DST_FULL, F.ACCUM_ADD, x, x.t(), -1.0))
        print(f"full  r={r} k={k}: {ms:.3f} ms {2 * r * r * k / ms / 1e9:.1f} TF", flush=True)
        ms = bench(lambda: F.gemm(c, F.DST_LOWER, F.ACCUM_ADD, x, x.t(), -1.0))
        nt = r // 128
        fl = nt * (nt + 1) / 2 * 128 * 128 * k * 2
        print(f"lower r={r} k={k}: {ms:.3f} ms {fl / ms / 1e9:.1f} TF (tile flops)  {nt * (nt + 1) // 2} tiles = {nt * (nt + 1) / 2 / 512:.2f} waves", flush=True)
        del x, c
