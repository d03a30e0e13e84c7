"""A/B of the accumulate epilogue (FAER_HIP_GEMM_EPI=0: one read-modify-write per element, 1: batched) on the
update shapes of the factorizations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()


def cm(m, n, dt):
    return torch.randn((n, m), dtype=dt, device="cuda").t()


def bench(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [(torch.float32, 500000, 8, 8), (torch.float32, 500000, 24, 8), (torch.float32, 500000, 64, 16), (torch.float32, 500000, 128, 128), (torch.float32, 500000, 64, 64), (torch.float32, 500000, 32, 32), (torch.float32, 500000, 16, 16),
          (torch.float64, 500000, 128, 128), (torch.float64, 16384, 256, 256), (torch.float64, 16384, 64, 64), (torch.float64, 16384, 448, 64),
          (torch.float64, 15360, 15360, 1024), (torch.float64, 15360, 15360, 512), (torch.float64, 4096, 4096, 128), (torch.float64, 2048, 2048, 2048),
          (torch.float32, 8192, 8192, 1024)]
if len(sys.argv) > 1 and sys.argv[1] == "skinny":
    shapes = [(torch.float32, 500000, 8, 8), (torch.float32, 500000, 16, 16), (torch.float32, 500000, 32, 32), (torch.float64, 500000, 16, 16),
              (torch.float32, 8, 8, 500000), (torch.float32, 16, 16, 500000), (torch.float64, 8, 8, 500000)]
    for dt, m, n, k in shapes:
        if k > 1000:
            a, b, c = cm(k, m, dt).t(), cm(k, n, dt), cm(m, n, dt)
        else:
            a, b, c = cm(m, k, dt), cm(k, n, dt), cm(m, n, dt)
        t = bench(lambda: F.matmul(c, F.ACCUM_ADD, a, b, -1.0), 20)
        byts = (m * k + k * n + 2 * m * n) * a.element_size()
        print(f"[{os.environ.get('FAER_HIP_SKINNY_R', '')},{os.environ.get('FAER_HIP_SKINNY_WGS', '')},{'off' if os.environ.get('FAER_HIP_NO_SKINNY') else 'on'}] {str(dt)[6:]:8s} {m:7d} x {n:5d} x {k:7d}: {t:8.1f} us  {byts / t / 1e6:6.2f} TB/s", flush=True)
    sys.exit(0)
for dt, m, n, k in shapes:
    a, b, c = cm(m, k, dt), cm(k, n, dt), cm(m, n, dt)
    res = []
    for epi in ("0", "1"):
        os.environ["FAER_HIP_GEMM_EPI"] = epi
        res.append(bench(lambda: F.matmul(c, F.ACCUM_ADD, a, b, -1.0)))
    print(f"{str(dt)[6:]:8s} {m:7d} x {n:5d} x {k:5d}: serial {res[0]:9.1f} us  batched {res[1]:9.1f} us  ({res[0] / res[1]:.2f}x)", flush=True)
    del a, b, c
