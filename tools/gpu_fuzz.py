"""Randomised sweep of the C-ABI on a GPU box: shapes, strides (Fortran / C / padded / reversed views), accumulate
modes, both dtypes -- GEMM, triangular products, triangular solves and the factorizations against float64 numpy
definitions.  Complements the fixed parametrisations of tests/ (run by hand: python tools/gpu_fuzz.py [cases])."""
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EPS = {np.float64: 2.3e-16, np.float32: 1.2e-7}
fails = 0


def dev_view(x):
    """device tensor holding x with a random memory layout"""
    m, n = x.shape
    t = torch.from_numpy(np.ascontiguousarray(x))
    kind = rng.integers(0, 5)
    if kind == 0:  # column major
        return t.t().contiguous().t().cuda()
    if kind == 1:  # row major
        return t.cuda()
    if kind == 2:  # padded column major
        big = torch.zeros((n, m + 3), dtype=t.dtype).cuda()
        big[:, :m] = t.t().cuda()
        return big[:, :m].t()
    if kind == 3:  # every other element of a bigger matrix
        big = torch.zeros((2 * m + 1, 2 * n + 1), dtype=t.dtype).cuda()
        big[::2, ::2][:m, :n] = t.cuda()
        return big[::2, ::2][:m, :n]
    big = torch.zeros((n + 2, m + 5), dtype=t.dtype).cuda()  # offset view of a padded column major matrix
    big[1:n + 1, 2:m + 2] = t.t().cuda()
    return big[1:n + 1, 2:m + 2].t()


def check(name, got, ref, tol, info):
    global fails
    err = np.abs(got.astype(np.float64) - ref).max(initial=0)
    if not np.isfinite(got).all() or err > tol:
        fails += 1
        print(f"FAIL {name} {info}: err {err:.3e} tol {tol:.3e}", flush=True)


def dims(hi):
    return int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 200, 257, hi]))


ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for case in range(ncases):
    dt = np.float64 if rng.integers(0, 2) else np.float32
    e = EPS[dt]
    which = rng.integers(0, 7)
    if which == 0:  # matmul incl. level-2 and tall-skinny shapes
        m, n, k = dims(600), dims(600), dims(600)
        if rng.integers(0, 6) == 0:
            m, n, k = [(20000, int(rng.integers(1, 33)), int(rng.integers(1, 17))), (int(rng.integers(1, 17)), int(rng.integers(1, 17)), 17000),
                       (int(rng.integers(1, 33)), 18000, int(rng.integers(1, 17)))][rng.integers(0, 3)]
        a, b, c = rng.standard_normal((m, k)).astype(dt), rng.standard_normal((k, n)).astype(dt), rng.standard_normal((m, n)).astype(dt)
        add, alpha = bool(rng.integers(0, 2)), float(rng.choice([1.0, -1.0, 0.5, 2.5]))
        dc = dev_view(c)
        F.matmul(dc, F.ACCUM_ADD if add else F.ACCUM_REPLACE, dev_view(a), dev_view(b), alpha)
        ref = alpha * (a.astype(np.float64) @ b.astype(np.float64)) + (c if add else 0)
        tol = 8 * max(k, 1) * e * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64) + np.abs(c)).max()
        check("matmul", dc.cpu().numpy(), ref, tol, (dt.__name__, m, n, k, add, alpha))
    elif which == 1:  # triangular solves
        n, k = dims(400), dims(300)
        t = (rng.standard_normal((n, n)) / max(n, 1) ** 0.5 + 2 * np.eye(n)).astype(dt)
        upper, unit = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        tri = np.triu(t) if upper else np.tril(t)
        if unit:
            np.fill_diagonal(tri, 1.0)
        b = rng.standard_normal((n, k)).astype(dt)
        x = dev_view(b)
        fn = {(False, False): F.solve_lower_triangular_in_place, (True, False): F.solve_upper_triangular_in_place,
              (False, True): F.solve_unit_lower_triangular_in_place, (True, True): F.solve_unit_upper_triangular_in_place}[(upper, unit)]
        fn(dev_view(t), x)
        ref = np.linalg.solve(tri.astype(np.float64), b.astype(np.float64))
        check("trsm", x.cpu().numpy(), ref, 64 * n * e * max(1.0, np.abs(ref).max()) * np.linalg.cond(tri.astype(np.float64)), (dt.__name__, n, k, upper, unit))
    elif which == 2:  # llt + solve
        n = dims(500)
        g = rng.standard_normal((n, n))
        a = (g @ g.T + n * np.eye(n)).astype(dt)
        l = dev_view(a)
        F.llt_factor_in_place(l)
        L = np.tril(l.cpu().numpy().astype(np.float64))
        check("llt", L @ L.T, a.astype(np.float64), 64 * n * e * np.abs(a).max(), (dt.__name__, n))
    elif which == 3:  # partial-pivot LU
        m, n = dims(500), dims(500)
        a = rng.standard_normal((m, n)).astype(dt)
        lu = dev_view(a)
        fwd, _, _ = F.partial_piv_lu_factor_in_place(lu)
        h = lu.cpu().numpy().astype(np.float64)
        s = min(m, n)
        check("lu", (np.tril(h[:, :s], -1) + np.eye(m, s)) @ np.triu(h[:s, :]), a[fwd.astype(int)].astype(np.float64), 64 * max(m, n) * e * np.abs(a).max(),
              (dt.__name__, m, n))
    elif which == 4:  # QR
        m, n = dims(600), dims(300)
        a = rng.standard_normal((m, n)).astype(dt)
        qr = dev_view(a)
        bs = F.qr_recommended_block_size(m, n, dt)
        hh = torch.zeros((min(m, n), bs), dtype=qr.dtype, device="cuda").t()
        F.qr_factor_in_place(qr, hh)
        out = torch.full((n, m), float("nan"), dtype=qr.dtype, device="cuda").t()
        F.qr_reconstruct(out, qr, hh)
        check("qr", out.cpu().numpy(), a.astype(np.float64), 64 * max(m, n) * e * np.abs(a).max(), (dt.__name__, m, n))
    elif which == 5:  # full-pivot LU
        m, n = dims(300), dims(300)
        a = rng.standard_normal((m, n)).astype(dt)
        lu = dev_view(a)
        rf, _, cf, _, _ = F.full_piv_lu_factor_in_place(lu)
        h = lu.cpu().numpy().astype(np.float64)
        s = min(m, n)
        check("fplu", (np.tril(h[:, :s], -1) + np.eye(m, s)) @ np.triu(h[:s, :]), a[rf.astype(int)][:, cf.astype(int)].astype(np.float64),
              64 * max(m, n) * e * np.abs(a).max(), (dt.__name__, m, n))
    else:  # triangular matmul with random structures
        n = dims(300)
        a, b, c = (rng.standard_normal((n, n)).astype(dt) for _ in range(3))
        names = ["rect", "lower", "upper", "strict_lower", "strict_upper", "unit_lower", "unit_upper"]
        sa, sb, sc = (names[rng.integers(0, 7)] for _ in range(3))

        def st(x, s):
            x = x.astype(np.float64)
            if s == "rect":
                return x
            t = np.tril(x, -1 if "strict" in s or "unit" in s else 0) if "lower" in s else np.triu(x, 1 if "strict" in s or "unit" in s else 0)
            return t + np.eye(n) if "unit" in s else t

        dc = dev_view(c)
        F.matmul_triangular(dc, sc, F.ACCUM_ADD, dev_view(a), sa, dev_view(b), sb, -0.5)
        prod = -0.5 * st(a, sa) @ st(b, sb)
        mask = np.ones((n, n), bool) if sc == "rect" else (np.tril(np.ones((n, n), bool), -1 if ("strict" in sc or "unit" in sc) else 0) if "lower" in sc
                                                      else np.triu(np.ones((n, n), bool), 1 if ("strict" in sc or "unit" in sc) else 0))
        ref = c.astype(np.float64) + np.where(mask, prod, 0.0)
        check("matmul_triangular", dc.cpu().numpy(), ref, 16 * n * e * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64) + np.abs(c) + 1).max(),
              (dt.__name__, n, sa, sb, sc))
F.synchronize()
print(f"fuzz: {ncases} cases, {fails} failures", flush=True)
sys.exit(1 if fails else 0)
