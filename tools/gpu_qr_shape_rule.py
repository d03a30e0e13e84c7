"""where the whole-matrix one-pass QR path beats the classic path (with its one-pass panels): moderately tall shapes, both dtypes"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
F = ge.load_package(); torch.cuda.set_device(0); F.lib(); F.use_torch_stream()
lib = F.lib()
import ctypes as C
lib.faer_hip_debug_qr_one_pass_shape_rule.argtypes = [C.c_long, C.c_long]
lib.faer_hip_debug_qr_one_pass_columns.restype = C.c_long
for dt in (torch.float64, torch.float32):
    for m, n in [(1024, 340), (1536, 512), (1100, 100), (2048, 64), (2048, 256), (2048, 512), (4096, 64), (4096, 256), (4096, 512), (8192, 64), (8192, 256), (8192, 512), (12000, 512), (16384, 256), (3000, 512), (1024, 128), (1024, 256)]:
        g = torch.Generator(device="cuda").manual_seed(1)
        a = torch.randn((n, m), dtype=dt, device="cuda", generator=g).t()
        bs = int(F.qr_recommended_block_size(m, n, "float64" if dt == torch.float64 else "float32"))
        res = []
        for rule in ((1 << 40, 8), (0, 0)):  # classic path forced / the default rule
            lib.faer_hip_debug_qr_one_pass_shape_rule(*rule)
            best = 1e9
            for rep in range(3):
                w = a.clone(); h = torch.zeros((n, bs), dtype=dt, device="cuda").t()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                F.qr_factor_in_place(w, h)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            res.append((best * 1e3, lib.faer_hip_debug_qr_one_pass_columns()))
        lib.faer_hip_debug_qr_one_pass_shape_rule(0, 0)
        print(f"{str(dt)[6:]} {m:6d} x {n:3d} bs {bs:3d}: classic (+ one-pass panels) {res[0][0]:7.2f} ms   whole-matrix one-pass {res[1][0]:7.2f} ms (columns {res[1][1]})", flush=True)
