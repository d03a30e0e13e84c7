"""GPU parity: GEMM / triangular product through the C-ABI vs the CPU oracle (SURVEY.md section 4:
test_matmul, test_triangular, the matmul doctests).  Tolerance: |d| <= 4 K eps (|A||B|)_ij."""
import json
import os

import numpy as np
import pytest

from gpu_util import EPS, fa, init_gpu, rnd, to_dev, to_host

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SHAPES = [(1, 1, 1), (15, 16, 17), (17, 15, 16), (127, 129, 128), (129, 127, 130), (64, 1, 33), (1, 70, 9), (5, 7, 0),
          (256, 256, 256), (300, 200, 1000), (40, 3, 5000), (130, 130, 70), (1, 1, 300), (513, 67, 1),
          (8, 1, 20000), (1, 6, 9000), (2000, 1, 4100), (700, 900, 1),  # level-2 shapes incl. split reductions
          # tall-skinny streams (skinny.hip): update, transposed update, split reduction; ragged blocks
          (20000, 8, 8), (20001, 30, 17), (16385, 32, 32), (17000, 3, 5), (8, 20000, 8), (32, 16400, 30),
          (8, 8, 20000), (16, 13, 16500), (5, 16, 17001)]


def bound(a, b, c0, k, dtype, alpha):
    e = EPS[np.dtype(dtype)]
    return 4 * max(k, 1) * e * (abs(alpha) * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)) +
                               np.abs(c0)) + 1e-300


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("layout", ["FFF", "CFF", "FCF", "CCF", "FFC", "CCC"])
@pytest.mark.parametrize("m,n,k", SHAPES)
def test_matmul_vs_oracle(oracle, dtype, layout, m, n, k):
    F = init_gpu()
    rng = np.random.default_rng(m * 1000003 + n * 1009 + k)
    a, b, c0 = rnd(rng, m, k, dtype), rnd(rng, k, n, dtype), rnd(rng, m, n, dtype)
    da, db = to_dev(a, layout[0]), to_dev(b, layout[1])
    for accum, alpha in ((F.ACCUM_ADD, -0.5), (F.ACCUM_REPLACE, 2.0)):
        dc = to_dev(c0 if accum == F.ACCUM_ADD else np.full((m, n), np.nan, dtype=dtype), layout[2])
        F.matmul(dc, accum, da, db, alpha)
        ref = c0.copy(order="F")
        oracle.matmul(ref, a, b, alpha=alpha, accum_add=accum == F.ACCUM_ADD)
        got = to_host(dc)
        tol = bound(a, b, c0 if accum == F.ACCUM_ADD else 0 * c0, k, dtype, alpha)
        assert np.isfinite(got).all()
        assert (np.abs(got.astype(np.float64) - ref.astype(np.float64)) <= tol).all(), \
            f"max err {np.abs(got - ref).max():.3e}"


def test_golden_matmul_2x2():
    F = init_gpu()
    g = json.load(open(os.path.join(GOLD, "matmul_2x2.json")))
    acc = to_dev(np.full((2, 2), np.nan))
    F.matmul(acc, F.ACCUM_REPLACE, to_dev(np.array(g["lhs"])), to_dev(np.array(g["rhs"])), g["alpha"])
    assert np.abs(to_host(acc) - np.array(g["target"])).max() < g["tol"]


def test_matmul_negative_and_generic_strides(oracle):
    F = init_gpu()
    rng = np.random.default_rng(5)
    a, b = rnd(rng, 70, 90), rnd(rng, 90, 50)
    big_a, big_b = to_dev(np.zeros((140, 180))), to_dev(np.zeros((180, 100)))
    big_a[::2, ::2] = to_dev(a)
    big_b[::2, ::2] = to_dev(b)
    va = big_a[::2, ::2].flip(0)  # generic (non unit) strides + reversed rows (torch has no negative strides)
    vb = big_b[::2, ::2]
    dc = to_dev(np.zeros((70, 50)))
    F.matmul(dc, F.ACCUM_REPLACE, va, vb, 1.0)
    assert np.abs(to_host(dc) - a[::-1] @ b).max() < 1e-11


def test_matmul_device_negative_strides_and_huge_leading_dimension(oracle):
    """operand views the buffer-addressed loaders of the pipelined kernel cannot express (gemm.hip, gemm_dev: negative device
    strides, per-tile offsets beyond 32 bits) are routed to the non-pipelined kernel: same results"""
    import ctypes as C
    F = init_gpu()
    rng = np.random.default_rng(15)
    a, b = rnd(rng, 200, 150), rnd(rng, 150, 130)
    da, db, dc = to_dev(a), to_dev(b), to_dev(np.zeros((200, 130)))
    # rows of a reversed, columns of b reversed, through hand-made views (torch has no negative strides)
    va = F.MatRef(da.data_ptr() + 199 * da.stride(0) * 8, 200, 150, -da.stride(0), da.stride(1))
    vb = F.MatRef(db.data_ptr() + 129 * db.stride(1) * 8, 150, 130, db.stride(0), -db.stride(1))
    al = C.c_double(1.0)
    F.lib().libfaer_v0_23_matmul_f64(F._mat(dc, F.MatMut), C.c_int(F.ACCUM_REPLACE), va, vb, C.byref(al), F.PAR_SEQ)
    assert np.abs(to_host(dc) - a[::-1] @ b[:, ::-1]).max() < 1e-11
    # K-major rhs with a leading dimension of 1.2e6 rows (the V^T A products of a very tall QR): C = X^T Y, K = 1.2e6
    import torch
    k = 1_200_000
    x = torch.randn((24, k), dtype=torch.float64, device="cuda").t()  # k x 24, column major
    y = torch.randn((20, k), dtype=torch.float64, device="cuda").t()
    c = torch.zeros((20, 24), dtype=torch.float64, device="cuda").t()  # 24 x 20, column major
    F.matmul(c, F.ACCUM_REPLACE, x.t(), y, 1.0)
    ref = x.t() @ y
    assert (c - ref).abs().max().item() <= 64 * k * 2.3e-16 * 16


@pytest.mark.parametrize("bcs", [0, 1, 3, 8, 15, 16, 17])
@pytest.mark.parametrize("shape", [(300, 260, 96), (128, 128, 64), (70, 33, 200)])
def test_matmul_rhs_broadcast_and_overlapping_columns(bcs, shape):
    """a K-major rhs whose columns are fewer than 16 elements apart -- a broadcast view (column stride 0: every column the
    same vector) or overlapping windows of one long vector (Hankel / sliding-window views): the descriptor-addressed B loader
    of the pipelined kernel ends where column N begins and would cut the last columns short, so gemm_dev routes
    them to the pointer loaders.  Also the edge tiles of an ordinary rhs placed at the very END of its allocation."""
    import ctypes as C
    import torch
    F = init_gpu()
    m, n, k = shape
    rng = np.random.default_rng(bcs * 7 + m)
    a = rnd(rng, m, k)
    vec = rng.standard_normal(k + max(bcs, 1) * n + 16)
    dvec = to_dev(vec.reshape(-1, 1))
    da, dc = to_dev(a), to_dev(np.full((m, n), np.nan))
    vb = F.MatRef(dvec.data_ptr(), k, n, 1, bcs)
    al = C.c_double(1.0)
    F.lib().libfaer_v0_23_matmul_f64(F._mat(dc, F.MatMut), C.c_int(F.ACCUM_REPLACE), F._mat(da), vb, C.byref(al), F.PAR_SEQ)
    b = np.stack([vec[j * bcs: j * bcs + k] for j in range(n)], axis=1)
    ref = a @ b
    assert np.abs(to_host(dc) - ref).max() <= 64 * k * 2.3e-16 * max(1.0, np.abs(ref).max())


def test_matmul_rhs_at_the_end_of_its_allocation():
    """edge tiles (N not a multiple of the tile) of a column-major rhs that ends exactly where its allocation ends: the scalar
    offsets of the B loader are clamped to the operand (the hardware checks only the per-lane part)"""
    import torch
    F = init_gpu()
    m, n, k = 512, 2 * 128 + 37, 512  # big tiles (>= 256 of them is not needed: the 64 x 64 pipelined tile takes the same loader)
    for nn in (n, 1000 + 1):
        a = torch.randn((k, m), dtype=torch.float64, device="cuda").t()
        b = torch.randn((nn, k), dtype=torch.float64, device="cuda").t()  # k x nn column major: its last column ends the allocation
        c = torch.empty((nn, m), dtype=torch.float64, device="cuda").t()
        F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0)
        F.synchronize()
        assert (c - a @ b).abs().max().item() <= 64 * k * 2.3e-16 * 16


def test_matmul_host_pointers(oracle):
    """host-resident operands (a faer::Mat): staged through device buffers by the library"""
    F = init_gpu()
    rng = np.random.default_rng(6)
    a, b, c0 = rnd(rng, 200, 150), rnd(rng, 150, 120), rnd(rng, 200, 120)
    c = c0.copy(order="F")
    F.matmul(c, F.ACCUM_ADD, a[::-1, :], b[:, ::-1], 1.5)  # negative host strides too
    assert np.abs(c - (c0 + 1.5 * a[::-1, :] @ b[:, ::-1])).max() < 1e-11


STRUCTS = ["rect", "lower", "upper", "strict_lower", "strict_upper", "unit_lower", "unit_upper"]


def dense_of(a, s):
    n = a.shape[0]
    return {"rect": a, "lower": np.tril(a), "upper": np.triu(a), "strict_lower": np.tril(a, -1),
            "strict_upper": np.triu(a, 1), "unit_lower": np.tril(a, -1) + np.eye(n),
            "unit_upper": np.triu(a, 1) + np.eye(n)}[s]


def mask_of(n, s):
    i, j = np.indices((n, n))
    return {"rect": i >= -1, "lower": i >= j, "upper": i <= j, "strict_lower": i > j, "strict_upper": i < j,
            "unit_lower": i > j, "unit_upper": i < j}[s]


@pytest.mark.parametrize("cs", STRUCTS)
@pytest.mark.parametrize("as_", STRUCTS)
def test_triangular_matmul(oracle, cs, as_):
    """matmul/mod.rs:2216-2266 `test_triangular`: all 7^3 structure triples, 1e-10, untouched part unchanged"""
    F = init_gpu()
    rng = np.random.default_rng(STRUCTS.index(cs) * 7 + STRUCTS.index(as_))
    for bs_ in STRUCTS:
        n = int(rng.integers(1, 100))
        a, b, c0 = rnd(rng, n, n), rnd(rng, n, n), rnd(rng, n, n)
        for accum in (F.ACCUM_ADD, F.ACCUM_REPLACE):
            dc = to_dev(c0)
            F.matmul_triangular(dc, cs, accum, to_dev(a), as_, to_dev(b), bs_, 2.5)
            got = to_host(dc)
            ref = c0.copy(order="F")
            oracle.matmul_triangular(ref, cs, a, as_, b, bs_, alpha=2.5, accum_add=accum == F.ACCUM_ADD)
            mk = mask_of(n, cs)
            assert np.abs(got - ref)[mk].max(initial=0) < 1e-10, (cs, as_, bs_, n)
            assert (got[~mk] == c0[~mk]).all(), (cs, as_, bs_, n)
            full = (c0 if accum == F.ACCUM_ADD else 0) + 2.5 * dense_of(a, as_) @ dense_of(b, bs_)
            assert np.abs(got - full)[mk].max(initial=0) < 1e-10


@pytest.mark.parametrize("kind", ["lower", "upper"])
@pytest.mark.parametrize("n,k", [(100, 37), (257, 128), (1000, 64)])
def test_gemm_inner_boundary_dst_kind(kind, n, k):
    """the private_gemm_x86::gemm(DstKind::Lower/Upper) call of triangular.rs:641-680"""
    F = init_gpu()
    rng = np.random.default_rng(n + k)
    a, c0 = rnd(rng, n, k), rnd(rng, n, n)
    dc = to_dev(c0)
    F.gemm(dc, F.DST_LOWER if kind == "lower" else F.DST_UPPER, F.ACCUM_ADD, to_dev(a), to_dev(a).t(), -1.0)
    got = to_host(dc)
    full = c0 - a @ a.T
    mk = np.tril(np.ones((n, n), bool)) if kind == "lower" else np.triu(np.ones((n, n), bool))
    assert np.abs(got - full)[mk].max() < 1e-10
    assert (got[~mk] == c0[~mk]).all()


def test_gemm_inner_boundary_scatter_and_diag():
    """row/col index scatter + fused diagonal scaling (matmul/internal/mod.rs:45-379 `spicy_matmul`)"""
    F = init_gpu()
    rng = np.random.default_rng(9)
    m, n, k = 50, 40, 30
    a, b, d = rnd(rng, m, k), rnd(rng, k, n), rng.standard_normal(k)
    c0 = rnd(rng, 80, 70)
    ri = rng.permutation(80)[:m].astype(np.uint64)
    ci = rng.permutation(70)[:n].astype(np.uint64)
    c = c0.copy(order="F")
    F.gemm(c, F.DST_FULL, F.ACCUM_ADD, a, b, 1.0, row_idx=ri, col_idx=ci, diag=d)  # host operands
    ref = c0.copy()
    ref[np.ix_(ri.astype(int), ci.astype(int))] += a @ np.diag(d) @ b
    assert np.abs(c - ref).max() < 1e-11


@pytest.mark.parametrize("itype", [np.uint32, np.uint64])
@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("kind", ["full", "lower"])
def test_gemm_inner_boundary_scatter_index_types(itype, where, kind):
    """spicy_matmul surface (matmul/internal/mod.rs:143-201): u32 AND u64 index arrays, host and device operands,
    Full and Lower dst (the sparse supernodal callers pass u32 indices and a triangular dst)"""
    import torch

    F = init_gpu()
    rng = np.random.default_rng(11)
    m = n = 150
    k = 70
    a, d = rnd(rng, m, k), rng.standard_normal(k)
    b = np.asfortranarray(a.T) if kind == "lower" else rnd(rng, k, n)
    c0 = rnd(rng, 400, 400)
    sel = np.sort(rng.permutation(400)[:m])  # increasing: the lower triangle of the scattered block stays lower
    ri = sel.astype(itype)
    ci = (sel if kind == "lower" else rng.permutation(400)[:n]).astype(itype)
    full = a @ np.diag(d) @ b
    ref = c0.copy()
    blk = ref[np.ix_(ri.astype(int), ci.astype(int))]
    blk += np.tril(full) if kind == "lower" else full
    ref[np.ix_(ri.astype(int), ci.astype(int))] = blk
    dk = F.DST_LOWER if kind == "lower" else F.DST_FULL
    if where == "host":
        c = c0.copy(order="F")
        F.gemm(c, dk, F.ACCUM_ADD, a, b, 1.0, row_idx=ri, col_idx=ci, diag=d)
        got = c
    else:
        dc = to_dev(c0)
        it = np.int32 if itype == np.uint32 else np.int64
        dri, dci = torch.from_numpy(ri.astype(it)).cuda(), torch.from_numpy(ci.astype(it)).cuda()
        F.gemm(dc, dk, F.ACCUM_ADD, to_dev(a), to_dev(b), 1.0, row_idx=dri, col_idx=dci, diag=torch.from_numpy(d).cuda())
        got = to_host(dc)
    assert np.abs(got - ref).max() < 1e-11
    untouched = np.ones_like(c0, bool)
    untouched[np.ix_(ri.astype(int), ci.astype(int))] = False
    assert (got[untouched] == c0[untouched]).all()


@pytest.mark.parametrize("layout", ["F_block", "C_block", "strided", "reversed"])
@pytest.mark.parametrize("accum", ["replace", "add"])
def test_host_submatrix_dst_leaves_the_parent_untouched(layout, accum):
    """a host dst that is a view of a larger matrix -- dst.submatrix_mut(..) of a parent Mat, the
    call pattern faer's own recursions produce -- must only have ITS elements written; the parent's entries between
    the view's columns are never read back or rewritten (with Replace they were staged as garbage in round 1)"""
    F = init_gpu()
    rng = np.random.default_rng(12)
    parent0 = rnd(rng, 60, 50, order="F" if layout != "C_block" else "C")
    parent = parent0.copy(order="K")
    view = {"F_block": parent[7:37, 5:25], "C_block": parent[7:37, 5:25], "strided": parent[3:57:2, 1:49:3],
            "reversed": parent[40:10:-1, 30:10:-1]}[layout]
    m, n = view.shape
    a, b = rnd(rng, m, 13), rnd(rng, 13, n)
    before = view.copy()
    mask = np.zeros(parent.shape, bool)
    pm = {"F_block": mask[7:37, 5:25], "C_block": mask[7:37, 5:25], "strided": mask[3:57:2, 1:49:3],
          "reversed": mask[40:10:-1, 30:10:-1]}[layout]
    pm[...] = True
    if accum == "replace":
        view[...] = np.nan  # Replace never reads dst
        F.matmul(view, F.ACCUM_REPLACE, a, b, 2.0)
        want = 2.0 * (a @ b)
    else:
        F.matmul(view, F.ACCUM_ADD, a, b, 2.0)
        want = before + 2.0 * (a @ b)
    assert np.abs(view - want).max() < 1e-12
    assert (parent[~mask] == parent0[~mask]).all()
    # the same through the inner boundary (faer_hip_gemm, Full / Replace) and an in-place factorization of a block
    view[...] = before
    F.gemm(view, F.DST_FULL, F.ACCUM_REPLACE, a, b, 1.0)
    assert np.abs(view - a @ b).max() < 1e-12 and (parent[~mask] == parent0[~mask]).all()
    if m == n:
        g = rng.standard_normal((m, m))
        view[...] = g @ g.T + m * np.eye(m)
        spd = view.copy()
        assert F.llt_factor_in_place(view) == 0
        L = np.tril(view)
        assert np.abs(L @ L.T - spd).max() < 1e-10 * np.abs(spd).max()
        assert (parent[~mask] == parent0[~mask]).all()


def test_matmul_full_size_property():
    """BASELINE config G (N = 8192 fp64): checksum (A B) x == A (B x), independent of the oracle"""
    import torch

    F = init_gpu()
    n = 8192
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    c = torch.empty((n, n), dtype=torch.float64, device="cuda").t()
    F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0)
    F.synchronize()
    x = torch.randn((n,), dtype=torch.float64, device="cuda", generator=g)
    lhs = c @ x
    rhs = a @ (b @ x)
    scale = (a.abs() @ (b.abs() @ x.abs())).max().item()
    assert (lhs - rhs).abs().max().item() <= 8 * n * 2.3e-16 * scale


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_matmul_config_r_1024_cubed_vs_oracle(oracle, dtype):
    """BASELINE config R (the reference's own CPU-runnable case: 1024^3 matmul) entry by entry against the oracle"""
    F = init_gpu()
    n = 1024
    rng = np.random.default_rng(1024)
    a, b, c0 = rnd(rng, n, n, dtype), rnd(rng, n, n, dtype), rnd(rng, n, n, dtype)
    for accum, alpha in ((F.ACCUM_ADD, -0.5), (F.ACCUM_REPLACE, 1.0)):
        dc = to_dev(c0 if accum == F.ACCUM_ADD else np.full((n, n), np.nan, dtype=dtype))
        F.matmul(dc, accum, to_dev(a), to_dev(b), alpha)
        ref = c0.copy(order="F")
        oracle.matmul(ref, a, b, alpha=alpha, accum_add=accum == F.ACCUM_ADD)
        got = to_host(dc)
        tol = bound(a, b, c0 if accum == F.ACCUM_ADD else 0 * c0, n, dtype, alpha)
        assert (np.abs(got.astype(np.float64) - ref.astype(np.float64)) <= tol).all()


def test_matmul_full_size_sampled_rows_vs_oracle(oracle):
    """BASELINE config G (N = 8192 fp64), the launch bench.py times: 96 sampled rows of C (all 8192 columns, i.e. every tile
    column and 96 tile rows of the 128 x 128 raster) against the oracle's product of the same rows, 4 K eps (|A||B|)"""
    import torch

    F = init_gpu()
    n = 8192
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    c = torch.empty((n, n), dtype=torch.float64, device="cuda").t()
    F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0)
    F.synchronize()
    rows = np.unique(np.concatenate([np.arange(0, n, 128)[:64] + np.arange(64) % 128, np.array([0, 1, 127, 128, 4095, 4096, n - 1]),
                                      np.random.default_rng(7).integers(0, n, 25)]))
    rt = torch.from_numpy(rows).cuda()
    ah = np.asfortranarray(a[rt].cpu().numpy())
    bh = np.asfortranarray(b.cpu().numpy())
    ref = np.zeros((len(rows), n), order="F")
    oracle.matmul(ref, ah, bh)
    got = c[rt].cpu().numpy()
    tol = 4 * n * EPS[np.dtype(np.float64)] * (np.abs(ah) @ np.abs(bh))
    assert (np.abs(got - ref) <= tol).all(), np.abs(got - ref).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind,m,n,k", [("full", 300, 500, 70), ("full", 128, 256, 16), ("full", 1000, 1030, 33), ("full", 257, 513, 129),
                                        ("lower", 1000, 1000, 90), ("lower", 129, 129, 40), ("lower", 640, 640, 16), ("lower", 1537, 1537, 70)])
def test_gemm_wide_tile_vs_oracle(oracle, kind, m, n, k, dtype):
    """the eight-wavefront 128 x 256 tile (gemm.hip: default for large plain products, incl. the trapezoid enumeration of a
    square lower dst) forced on small ragged shapes: entries within 4 K eps (|A||B|), the strict upper triangle of a lower
    dst untouched bit for bit"""
    F = init_gpu()
    rng = np.random.default_rng(m + n + k)
    a, b, c0 = rnd(rng, m, k, dtype), rnd(rng, k, n, dtype), rnd(rng, m, n, dtype)
    F.lib().faer_hip_set_gemm_variant(3)
    try:
        dc = to_dev(c0)
        F.gemm(dc, F.DST_LOWER if kind == "lower" else F.DST_FULL, F.ACCUM_ADD, to_dev(a), to_dev(b), -0.75)
        got = to_host(dc)
    finally:
        F.lib().faer_hip_set_gemm_variant(0)
    ref = c0.copy(order="F")
    oracle.matmul(ref, a, b, alpha=-0.75, accum_add=True)
    tol = bound(a, b, c0, k, dtype, -0.75)
    if kind == "lower":
        lo = np.tril(np.ones((m, n), bool))
        assert (np.abs(got.astype(np.float64) - ref.astype(np.float64))[lo] <= tol[lo]).all()
        assert np.array_equal(got[~lo], c0[~lo])
    else:
        assert (np.abs(got.astype(np.float64) - ref.astype(np.float64)) <= tol).all()
