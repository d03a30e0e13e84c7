"""fuzz of qr_factor_in_place over the dispatch of the end of round 6 (whole-matrix one-pass path from 1024 rows, one-pass panels inside the
classic recursion, rebuilt T blocks, fp64 alignment rules): random shapes, block sizes, dtypes and VIEWS (row offset into a taller parent,
padded / odd column strides) against the oracle.  usage: gpu_qr_fuzz.py [cases] [seed]"""
import ctypes as C
import os
import sys

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
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
paths = {}
for it in range(cases):
    dtype = np.float64 if rng.random() < 0.5 else np.float32
    n = int(rng.choice([1, 7, 16, 33, 64, 65, 100, 128, 200, 256, 300, 512, 600]))
    kind = rng.random()
    m = int(n * rng.uniform(1.0, 2.5)) if kind < 0.3 else int(rng.integers(max(n, 256), 6000))
    if rng.random() < 0.1:
        m, n = n, m  # wide
        n = min(n, 1500)
    size = min(m, n)
    bs = rng.choice([1, 8, 15, 16, 32, 48, 64, 128, 0])
    bs = int(bs) if bs else int(F.qr_recommended_block_size(m, n, dtype))
    bs = max(1, min(bs, size))
    roff = int(rng.choice([0, 0, 1, 2, 3, 4, 16]))
    pad = int(rng.choice([0, 0, 1, 2, 3, 4, 8, 13]))
    ld = m + roff + pad
    a = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype))
    if rng.random() < 0.15 and n >= 3:  # a dependent column somewhere
        j = int(rng.integers(1, n))
        a[:, j] = a[:, :j] @ rng.standard_normal(j).astype(dtype) / np.sqrt(j)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    buf = torch.zeros((n, ld), dtype=tdt, device="cuda")
    buf[:, roff:roff + m] = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    dqr = buf.t()[roff:roff + m, :]
    dh = torch.zeros((size, bs), dtype=tdt, device="cuda").t()
    rank = F.qr_factor_in_place(dqr, dh)
    cols = lib.faer_hip_debug_qr_one_pass_columns()
    ref, rh = a.copy(order="F"), np.zeros((bs, size), dtype=dtype, order="F")
    rk = oracle.qr_in_place(ref, rh)
    F.synchronize()
    qr, h = dqr.cpu().numpy().astype(np.float64), dh.cpu().numpy().astype(np.float64)
    e = float(np.finfo(dtype).eps)
    tol = 64 * max(m, n) * e * max(1.0, np.abs(a).max())
    ok = rank == rk
    msg = ""
    if ok and rk == size:
        d = np.abs(qr - ref).max()
        fin = np.isfinite(rh)
        up = np.zeros((bs, size), bool)
        for j0 in range(0, size, bs):
            w = min(bs, size - j0)
            up[:w, j0:j0 + w] = np.triu(np.ones((w, w), bool))
        okh = (np.isfinite(h) == fin).all()
        dt_ = np.abs(h - np.where(fin, rh, 0.0))[fin & up].max(initial=0) / max(1.0, np.abs(rh[fin & up]).max(initial=0))
        ok = d <= 8 * tol and okh and dt_ <= 8 * tol
        msg = f"dQR {d / tol:.3f} tol  dT {dt_ / tol:.3f} tol"
        # nothing outside the view was touched
        ok = ok and float(buf[:, :roff].abs().sum()) == 0.0 and float(buf[:, roff + m:].abs().sum()) == 0.0
    elif ok:
        okh = np.array_equal(np.isinf(h), np.isinf(rh))
        ok = okh
        msg = f"rank {rk} of {size}"
    key = "one-pass" if cols == size else ("partial" if cols >= 0 else "classic")
    paths[key] = paths.get(key, 0) + 1
    if not ok:
        bad += 1
    print(f"{'ok ' if ok else 'BAD'} {np.dtype(dtype).name} {m} x {n} bs {bs} roff {roff} ld {ld}: rank {rank}/{rk} cols {cols} {msg}", flush=True)
print(f"{cases} cases, {bad} bad, paths {paths}")
sys.exit(1 if bad else 0)
