"""tall matrices with more than 512 columns (classic path with one-pass panels / nodes): times and rates"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
F = ge.load_package(); torch.cuda.set_device(0); F.lib(); F.use_torch_stream()
for dt in (torch.float64, torch.float32):
    for m, n in [(100000, 1024), (50000, 600), (20000, 2000), (200000, 768), (16384, 16384)]:
        if m == n and dt == torch.float32:
            continue
        g = torch.Generator(device="cuda").manual_seed(1)
        a = torch.randn((n, m), dtype=dt, device="cuda", generator=g).t()
        bs = int(F.qr_recommended_block_size(m, n, "float64" if dt == torch.float64 else "float32"))
        best = 1e9
        for rep in range(3):
            w = a.clone(); h = torch.zeros((n, bs), dtype=dt, device="cuda").t()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            F.qr_factor_in_place(w, h)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        fl = 2.0 * m * n * n - 2.0 / 3.0 * n ** 3
        print(f"{str(dt)[6:]} {m} x {n} bs {bs}: {best * 1e3:.2f} ms, {fl / best / 1e12:.2f} TFLOP/s", flush=True)
