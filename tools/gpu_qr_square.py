"""square / moderately tall Householder QR (classic path) timing, with and without the one-pass panels of csrc/tsqr.hip inside the
recursion: usage gpu_qr_square.py [n ...]   (both dtypes; errors of the factors against each other in eps)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
F = ge.load_package(); torch.cuda.set_device(0); F.lib(); F.use_torch_stream()
lib = F.lib()
sizes = [int(x) for x in sys.argv[1:]] or [2048, 4096, 8192]
for dt in (torch.float64, torch.float32):
    for n in sizes:
        g = torch.Generator(device="cuda").manual_seed(1)
        a = torch.randn((n, n), dtype=dt, device="cuda", generator=g).t()
        bs = int(F.qr_recommended_block_size(n, n, "float64" if dt == torch.float64 else "float32"))
        res = {}
        for on in (0, 1):
            lib.faer_hip_debug_qr_panels_one_pass(on)
            best = 1e9
            for rep in range(3):
                w = a.clone(); h = torch.zeros((n, bs), dtype=dt, device="cuda").t()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                rank = F.qr_factor_in_place(w, h)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            res[on] = (best, w.cpu().numpy().astype(np.float64), h.cpu().numpy().astype(np.float64), rank)
        lib.faer_hip_debug_qr_panels_one_pass(1)
        e = float(np.finfo(np.float64 if dt == torch.float64 else np.float32).eps)
        d = np.abs(res[0][1] - res[1][1]).max() / np.abs(res[0][1]).max() / e
        dh = np.abs(res[0][2] - res[1][2]).max() / np.abs(res[0][2]).max() / e
        print(f"qr {n} x {n} {str(dt)[6:]} bs {bs}: recursion only {res[0][0] * 1e3:8.2f} ms, one-pass panels {res[1][0] * 1e3:8.2f} ms "
              f"({4 / 3 * n ** 3 / res[1][0] / 1e12:.2f} TFLOP/s), ranks {res[0][3]} {res[1][3]}, factors differ by {d:.1f} eps (QR) {dh:.1f} eps (T) of the largest entry", flush=True)
