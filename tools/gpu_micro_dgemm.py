"""DGEMM N = 8192 (replace) and the K = 4096 accumulate product on the 128 x 256 tile: A/B of library builds.  GPU box only."""
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


def bench(fn, reps=8):
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
for _ in range(2):
    ms = bench(lambda: F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0))
    print(f"dgemm {n}^3 replace: {ms:.3f} ms {2 * n ** 3 / ms / 1e9:.1f} TF", flush=True)
