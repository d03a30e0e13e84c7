"""fp64 tall-skinny QR on the one-pass path (csrc/tsqr.hip, tsqr_factor64) against the classic path and the oracle:
per shape the columns the one-pass path completed, the errors of R / V / T against the oracle in fp64 eps (both paths)
and the times of both paths.  usage: gpu_qr_f64_onepass.py [big]   (big: also 5e5 x 256 with its oracle run, ~1 min of CPU)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

F = ge.load_package()
torch.cuda.set_device(0)
F.lib()
F.use_torch_stream()
from oracle import oracle  # noqa: E402  (checker of this tool only)

lib = F.lib()
lib.faer_hip_debug_qr_one_pass_columns.restype = C.c_long
E = float(np.finfo(np.float64).eps)


def errs(qr, h, ref, rh, bs):
    m, n = ref.shape
    up = np.triu(np.ones((m, n), bool))
    d = np.abs(qr - ref)
    dr = (np.where(up, d, 0.0).max(axis=0) / np.where(up, np.abs(ref), 0.0).max(axis=0)).max()
    dv = d[~up].max()
    tu = np.zeros((bs, n), bool)
    for j0 in range(0, n, bs):
        w = min(bs, n - j0)
        tu[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
    dt = np.abs(h - rh)[tu].max() / np.abs(rh[tu]).max()
    return dr / E, dv / E, dt / E


def run(a, bs, one_pass, reps=3):
    m, n = a.shape
    lib.faer_hip_debug_qr_one_pass_f64(1 if one_pass else 0)
    best = 1e9
    for _ in range(reps):
        w = a.clone()
        h = torch.zeros((n, bs), dtype=torch.float64, device="cuda").t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rank = F.qr_factor_in_place(w, h)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    cols = lib.faer_hip_debug_qr_one_pass_columns()
    lib.faer_hip_debug_qr_one_pass_f64(1)
    return w.cpu().numpy(), h.cpu().numpy(), rank, cols, best * 1e3


shapes = [(20000, 64, 64), (30000, 128, 128), (40000, 256, 256), (33000, 384, 128), (20000, 512, 512), (36000, 448, 64), (30001, 300, 64), (65536, 256, 32)]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    shapes.append((500000, 256, 256))
for m, n, bs in shapes:
    g = torch.Generator(device="cuda").manual_seed(m + n)
    a = torch.randn((n, m), dtype=torch.float64, device="cuda", generator=g).t()
    ah = a.cpu().numpy()
    ref, rh = np.asfortranarray(ah.copy()), np.zeros((bs, n), order="F")
    t0 = time.perf_counter()
    assert oracle.qr_in_place(ref, rh) == n
    tor = time.perf_counter() - t0
    out = []
    for one in (True, False):
        qr, h, rank, cols, ms = run(a, bs, one)
        out.append((one, rank, cols, ms) + errs(qr, h, ref, rh, bs))
    for one, rank, cols, ms, dr, dv, dt in out:
        print(f"{m:7d} x {n:3d} bs {bs:3d} {'one-pass' if one else 'classic '}: rank {rank} one-pass cols {cols:4d} {ms:8.2f} ms | vs oracle: R {dr:8.1f} eps  V {dv:8.2f} eps  T {dt:8.1f} eps"
              f"{'   (oracle %.1f s)' % tor if one else ''}", flush=True)
