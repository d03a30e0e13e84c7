"""diagnostic of the one-pass QR: R / V / T against the oracle in units of fp32 eps, per shape"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import __graft_entry__ as ge
from oracle import oracle

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
e = float(np.finfo(np.float32).eps)
for (m, n, bs) in [(20000, 64, 64), (20000, 64, 1), (20000, 128, 64), (20000, 128, 128), (30000, 256, 256)]:
    rng = np.random.default_rng(m + n)
    a = np.asarray(rng.standard_normal((m, n)), dtype=np.float32, order="F")
    for rep in range(2):
        dqr = torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()
        dh = torch.zeros((n, bs), dtype=torch.float32, device="cuda").t()
        rank = F.qr_factor_in_place(dqr, dh)
        F.synchronize()
        qr, h = dqr.cpu().numpy(), dh.cpu().numpy()
        ref, rh = a.copy(order="F"), np.zeros((bs, n), dtype=np.float32, order="F")
        oracle.qr_in_place(ref, rh)
        up = np.triu(np.ones((m, n), bool))
        d = np.abs(qr.astype(np.float64) - ref)
        tu = np.zeros((bs, n), bool)
        for j0 in range(0, n, bs):
            w = min(bs, n - j0)
            tu[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
        dT = np.abs(h.astype(np.float64) - rh)
        dTd = dT[tu].max()
        iT = np.unravel_index(np.argmax(np.where(tu, dT, 0)), dT.shape)
        iV = np.unravel_index(np.argmax(np.where(~up, d, 0)), d.shape)
        print(f"{m}x{n} bs={bs} rep{rep} rank={rank}: R {d[up].max() / e / np.abs(ref[up]).max():.2f} eps, V {d[~up].max() / e:.2f} eps at {iV}, "
              f"T {dTd / e:.2f} eps at {iT}", flush=True)
