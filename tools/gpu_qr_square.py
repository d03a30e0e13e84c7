"""square Householder QR (classic path) timing: usage gpu_qr_square.py [n] [f32|f64]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
F = ge.load_package(); torch.cuda.set_device(0); F.lib(); F.use_torch_stream()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn((n, n), dtype=dt, device="cuda", generator=g).t()
bs = int(F.qr_recommended_block_size(n, n, "float64" if dt == torch.float64 else "float32"))
best = 1e9
for rep in range(3):
    w = a.clone(); h = torch.zeros((n, bs), dtype=dt, device="cuda").t()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    F.qr_factor_in_place(w, h)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"qr {n} x {n} {dt}: {best * 1e3:.2f} ms (block size {bs}), {4 / 3 * n ** 3 / best / 1e12:.2f} TFLOP/s")
