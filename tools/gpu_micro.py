"""Launch-latency microbenchmarks of the small GEMM shapes the factorizations issue (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()


def cm(m, n, dtype=torch.float64):
    return torch.randn((n, m), dtype=dtype, device="cuda").t()


def bench(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


for variant in (0, 12):
    L.faer_hip_set_gemm_variant(variant)
    for (m, n, k) in [(64, 64, 16), (64, 64, 128), (64, 64, 1024), (512, 128, 128), (512, 512, 128), (512, 512, 512),
                      (2048, 2048, 128), (128, 128, 128)]:
        a, b, c = cm(m, k), cm(k, n), cm(m, n)
        t = bench(lambda: F.matmul(c, F.ACCUM_ADD, a, b, -1.0))
        print(f"variant {variant:2d} dgemm add {m}x{n}x{k}: {t:7.2f} us/launch", flush=True)
L.faer_hip_set_gemm_variant(0)
# alternating two different kernels (f64 / f32) vs the same kernel: cold instruction cache?
a, b, c = cm(512, 128), cm(128, 128), cm(512, 128)
a32, b32, c32 = cm(512, 128, torch.float32), cm(128, 128, torch.float32), cm(512, 128, torch.float32)
t_same = bench(lambda: (F.matmul(c, F.ACCUM_ADD, a, b, -1.0), F.matmul(c, F.ACCUM_ADD, a, b, -1.0)))
t_alt = bench(lambda: (F.matmul(c, F.ACCUM_ADD, a, b, -1.0), F.matmul(c32, F.ACCUM_ADD, a32, b32, -1.0)))
print(f"pair same kernel: {t_same:.2f} us ; pair alternating f64/f32 kernels: {t_alt:.2f} us")
# torch elementwise kernel as a launch-overhead yardstick
x = torch.zeros(1024, device="cuda")
print(f"torch add_ (tiny kernel): {bench(lambda: x.add_(1.0)):.2f} us/launch")
# in-place TRSM leaf product through the public TRSM API: 128 x 128 triangle, many rhs
l = torch.tril(cm(128, 128)) + 128 * torch.eye(128, dtype=torch.float64, device="cuda")
l = l.t().contiguous().t()
for k in (256, 2048, 16384):
    x = cm(128, k)
    t = bench(lambda: F.solve_lower_triangular_in_place(l, x), 50)
    print(f"trsm 128 x {k} rhs (trtri + in-place product): {t:.2f} us/call")
