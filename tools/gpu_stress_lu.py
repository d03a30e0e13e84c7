"""Determinism / race stress of the look-ahead LU and the QR panel: same factorization repeated with unrelated
work interleaved, compared BITWISE with the first result (pivots included)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cuda").manual_seed(4)
a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
big = torch.randn((6144, 6144), dtype=torch.float64, device="cuda")
ref = None
bad = 0
for it in range(8):
    work = a.clone()
    if it % 2 == 1:
        _ = big @ big
    perm, _, _ = F.partial_piv_lu_factor_in_place(work)
    F.synchronize()
    if ref is None:
        ref, ref_perm = work.clone(), perm.copy()
        p = torch.as_tensor(perm.astype(np.int64), device="cuda")
        L = torch.tril(work, -1) + torch.eye(n, dtype=torch.float64, device="cuda")
        x = torch.randn((n, 2), dtype=torch.float64, device="cuda")
        r = (L @ (torch.triu(work) @ x) - a[p] @ x).abs().max().item()
        print(f"lu iter 0: residual {r:.2e}")
    elif not (np.array_equal(perm, ref_perm) and torch.equal(work, ref)):
        print(f"lu iter {it}: MISMATCH (pivots equal: {np.array_equal(perm, ref_perm)})")
        bad += 1
# QR tall skinny fp32
m, nc = 500000, 256
aq = torch.randn((nc, m), dtype=torch.float32, device="cuda", generator=g).t()
refq = None
for it in range(6):
    w = aq.clone()
    h = torch.zeros((nc, nc), dtype=torch.float32, device="cuda").t()
    if it % 2 == 1:
        _ = big @ big
    rank = F.qr_factor_in_place(w, h)
    F.synchronize()
    if refq is None:
        refq, refh = w.clone(), h.clone()
        print("qr iter 0: rank", rank)
    elif not (torch.equal(w, refq) and torch.equal(h, refh)):
        d = (w != refq)
        print(f"qr iter {it}: MISMATCH in {int(d.sum().item())} entries of QR (split-K atomics make the GEMM sums order dependent)")
print("stress done, bad lu iterations:", bad)
