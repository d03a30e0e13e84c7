"""Randomised shapes for the one-pass tall-skinny QR (csrc/tsqr.hip) against the oracle: heights 16384 .. 90000, widths
1 .. 512 (ragged last panels of 1 .. 63 columns), every admissible block size of Q_coeff, leading dimensions and base
pointers that are / are not 16-byte aligned.  Usage: python tests/diag/gpu_qr_tall_fuzz.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import __graft_entry__ as ge
from oracle import oracle

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
F.lib().faer_hip_debug_qr_one_pass_columns.restype = C.c_long
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
e = float(np.finfo(np.float32).eps)
t_end = time.time() + budget
cases = worst_r = worst_v = worst_t = 0
while time.time() < t_end:
    n = int(rng.choice([rng.integers(1, 513), rng.choice([1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 320, 448, 511, 512])]))
    m = int(rng.integers(max(16384, 8 * n), 90001))
    divs = [b for b in (1, 2, 4, 8, 16, 32, 64, 128, 192, 256, 320, 384, 448, 512) if b <= max(n, 1) or b == 64]
    bs = int(rng.choice([b for b in divs if b % 64 == 0 or 64 % b == 0]))
    bs = max(1, min(bs, n)) if (min(bs, n) % 64 == 0 or 64 % min(bs, n) == 0) else 64 if n >= 64 else 1
    pad, off = int(rng.choice([0, 0, 1, 3, 4, 8])), int(rng.choice([0, 0, 1, 2, 4]))
    a = rng.standard_normal((m, n)).astype(np.float32)
    if rng.random() < 0.3:
        a *= np.exp(rng.uniform(-3, 3, n)).astype(np.float32)[None, :]  # column scales over a few decades
    ld = m + pad + off
    buf = torch.zeros((n, ld), dtype=torch.float32, device="cuda")
    buf[:, off:off + m] = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    dqr = buf.t()[off:off + m, :]
    dh = torch.zeros((n, bs), dtype=torch.float32, device="cuda").t()
    rank = F.qr_factor_in_place(dqr, dh)
    onep = F.lib().faer_hip_debug_qr_one_pass_columns()
    ref, rh = a.copy(order="F"), np.zeros((bs, n), dtype=np.float32, order="F")
    rk = oracle.qr_in_place(ref, rh)
    assert rank == rk == n, (m, n, bs, rank, rk)
    qr, h = dqr.cpu().numpy(), dh.cpu().numpy()
    up = np.triu(np.ones((m, n), bool))
    d = np.abs(qr.astype(np.float64) - ref)
    er, ev = d[up].max() / np.abs(ref[up]).max() / e, d[~up].max() / e
    tu = np.zeros((bs, n), bool)
    for j0 in range(0, n, bs):
        w = min(bs, n - j0)
        tu[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
    et = np.abs(h.astype(np.float64) - rh)[tu].max() / np.abs(rh[tu]).max() / e
    cases += 1
    worst_r, worst_v, worst_t = max(worst_r, er), max(worst_v, ev), max(worst_t, et)
    flag = "" if (er <= 64 and ev <= 16 and et <= 64 and np.isfinite(h).all()) else "  <-- OUT OF TOLERANCE"
    print(f"{m}x{n} bs={bs} pad={pad} off={off} one-pass columns {onep}: R {er:.2f} V {ev:.2f} T {et:.2f} eps{flag}", flush=True)
    assert not flag
print(f"{cases} cases, worst R {worst_r:.2f} V {worst_v:.2f} T {worst_t:.2f} eps")
