"""fp32 tall-skinny QR: the schedules of the one-pass path (faer_hip_debug_qr_fused: 1 fused default, 0 separate launches of rounds 3-5,
3 plain schedule on the streaming kernels of the end of round 6) -- time and agreement of the factors"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
F = ge.load_package(); torch.cuda.set_device(0); F.lib(); F.use_torch_stream()
lib = F.lib()
import ctypes as C
lib.faer_hip_debug_qr_one_pass_columns.restype = C.c_long
for m, n in [(500000, 256), (524287, 256), (100000, 512), (200000, 128), (30000, 256), (4096, 256)]:
    ldp = (m + 15) // 16 * 16
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn((n, ldp), dtype=torch.float32, device="cuda", generator=g)[:, :m].t()
    bs = int(F.qr_recommended_block_size(m, n, "float32"))
    res = {}
    for mode in (1, 0, 3):
        lib.faer_hip_debug_qr_fused(mode)
        best = 1e9
        for rep in range(5):
            w = torch.empty((n, ldp), dtype=torch.float32, device="cuda")[:, :m].t(); w.copy_(a)
            h = torch.zeros((n, bs), dtype=torch.float32, device="cuda").t()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            F.qr_factor_in_place(w, h)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res[mode] = (best * 1e3, w.cpu().numpy().astype(np.float64), h.cpu().numpy().astype(np.float64), lib.faer_hip_debug_qr_one_pass_columns())
    lib.faer_hip_debug_qr_fused(1)
    e = float(np.finfo(np.float32).eps)
    up = np.triu(np.ones((m, n), bool))
    def diff(x, y):
        d = np.abs(x - y)
        return (np.where(up, d, 0).max(axis=0) / np.where(up, np.abs(y), 0).max(axis=0)).max() / e, d[~up].max() / e
    d3 = diff(res[3][1], res[1][1]); dt = np.abs(res[3][2] - res[1][2]).max() / np.abs(res[1][2]).max() / e
    print(f"{m} x {n} bs {bs}: fused {res[1][0]:.3f} ms  separate (r3-5) {res[0][0]:.3f} ms  plain new kernels {res[3][0]:.3f} ms  (cols {res[1][3]} {res[0][3]} {res[3][3]}); "
          f"plain vs fused: R {d3[0]:.1f} eps V {d3[1]:.2f} eps T {dt:.1f} eps", flush=True)
