"""block-wise accuracy of a two-panel one-pass QR against the oracle (fp32 eps)"""
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
m, n, bs = 20000, 128, 64
rng = np.random.default_rng(m + n)
a = np.asarray(rng.standard_normal((m, n)), dtype=np.float32, order="F")
dqr = torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()
dh = torch.zeros((n, bs), dtype=torch.float32, device="cuda").t()
F.qr_factor_in_place(dqr, dh)
F.synchronize()
qr = dqr.cpu().numpy().astype(np.float64)
ref64 = a.astype(np.float64).copy(order="F")
rh = np.zeros((bs, n), order="F")
oracle.qr_in_place(ref64, rh)
d = np.abs(qr - ref64)
sc = np.abs(np.triu(ref64[:n])).max()
print("R11", np.triu(d[:64, :64]).max() / e / sc, "R12", d[:64, 64:].max() / e / sc, "R22", np.triu(d[64:128, 64:]).max() / e / sc)
print("V1 top", np.tril(d[:64, :64], -1).max() / e, "V1 below", d[64:, :64].max() / e, "V2 top", np.tril(d[64:128, 64:], -1).max() / e, "V2 below", d[128:, 64:].max() / e)
print("R12 per row max", (d[:64, 64:].max(axis=1) / e / sc).round(1)[::8], "R12 per col max", (d[:64, 64:].max(axis=0) / e / sc).round(1)[::8])
