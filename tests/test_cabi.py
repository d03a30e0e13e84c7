"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
declared in include/faer_hip.h, and its repr(C) structs have the layout of faer-ffi's (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from gpu_util import ROOT, fa


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "faer_hip.h")).read()
    hdr = "\n".join(l for l in hdr.splitlines() if not l.startswith("#define"))
    return re.findall(r"FAER_HIP_API\s+[^;(]*?\b(\w+)\s*\(", hdr)


def test_library_loads_and_exports_every_declared_symbol():
    F = fa()
    lib = F.lib()
    syms = declared_symbols()
    assert len(syms) >= 60
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert b"gfx950" in lib.faer_hip_version()


def test_struct_layouts_match_faer_ffi():
    F = fa()
    # faer-ffi/faer.h:203-217 MatRef/MatMut: ptr, nrows, ncols, row_stride, col_stride (5 x 8 bytes)
    assert C.sizeof(F.MatRef) == 40 and C.sizeof(F.MatMut) == 40
    assert [f[0] for f in F.MatRef._fields_] == ["ptr", "nrows", "ncols", "row_stride", "col_stride"]
    assert C.sizeof(F.SliceMut) == 16 and C.sizeof(F.Par) == 16 and C.sizeof(F.MemAlloc) == 16
    assert C.sizeof(F.LltStatus) == 16 and F.LltStatus.value.offset == 8  # tag + union
    assert C.sizeof(F.PartialPivLuStatus) == 16 and C.sizeof(F.QrStatus) == 16


def test_params_and_scratch_queries_need_no_gpu():
    F = fa()
    lib = F.lib()
    p = lib.libfaer_v0_23_LltParams_f64()
    assert (p.recursion_threshold, p.block_size) == (64, 128)  # cholesky/ldlt/factor.rs:705-714
    p = lib.libfaer_v0_23_PartialPivLuParams_f64()
    assert (p.recursion_threshold, p.block_size, p.par_threshold) == (16, 64, 128 * 128)
    p = lib.libfaer_v0_23_QrParams_f32()
    assert (p.blocking_threshold, p.par_threshold) == (48 * 48, 192 * 256)
    lay = lib.libfaer_v0_23_llt_factor_in_place_scratch_f64(C.c_size_t(100), F.PAR_SEQ, F.LltParams(64, 128))
    assert lay.len_bytes == 800
    lay = lib.libfaer_v0_23_qr_factor_in_place_scratch_f32(C.c_size_t(1000), C.c_size_t(50), C.c_size_t(8), F.PAR_SEQ,
                                                         F.QrParams(48 * 48, 192 * 256))
    assert lay.len_bytes == 8 * 50 * 4
    # qr/no_pivoting/factor.rs:91-116
    for (m, n, want) in [(10, 2, 1), (100, 100, 8), (600, 600, 48), (1000000, 256, 256), (3000, 3000, 128), (5, 100, 4), (3, 1000, 3)]:
        assert lib.libfaer_v0_23_qr_recommended_block_size_f64(C.c_size_t(m), C.c_size_t(n)) == want
    lib.libfaer_v0_23_set_global_par(F.Par(1, 8))
    g = lib.libfaer_v0_23_get_global_par()
    assert (g.tag, g.nthreads) == (1, 8)
    assert lib.faer_hip_dist_local_ncols(C.c_size_t(1000), C.c_size_t(128), 1, 4) == 256
    assert lib.faer_hip_dist_local_ncols(C.c_size_t(1000), C.c_size_t(128), 3, 4) == 232


def test_product_path_has_no_cpu_fallback():
    """without a gfx950 device a compute entry point must fail loudly, not compute on the host"""
    import subprocess
    import sys

    F = fa()
    if F.lib().faer_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, __graft_entry__ as g; F = g.load_package();"
            "a = np.eye(4, order='F'); c = np.zeros((4, 4), order='F'); F.matmul(c, 0, a, a, 1.0); print('computed')"
            % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "computed" not in r.stdout
    assert "no HIP device" in r.stderr or "faer_hip: fatal" in r.stderr


def test_driver_planning_logic_needs_no_gpu():
    """host-side decisions of the device drivers (pure functions, csrc/potrf.hip llt_plan, csrc/getrf.hip
    lu_leaf_width): step plan of the blocked Cholesky and the leaf shape of the cooperative LU panel"""
    F = fa()
    lib = F.lib()
    lib.faer_hip_debug_llt_plan.restype = C.c_size_t

    def plan(n, tail, nb2=1024):
        buf = (C.c_size_t * 64)()
        cnt = lib.faer_hip_debug_llt_plan(C.c_size_t(n), C.c_size_t(tail), C.c_size_t(nb2), buf, C.c_size_t(64))
        return [int(buf[i]) for i in range(cnt)]

    # N = 16384, look-ahead until at most 4096 rows remain: the first step is ONE 128-block (the whole chip waits for the
    # first diagonal block), the second 512 wide (its chain only has the K = 128 update to hide behind), then steps of 1024;
    # tail from column 12928
    assert plan(16384, 4096) == [0, 128] + [640 + 1024 * i for i in range(13)]
    # everything to the tail when the matrix is not larger than the tail threshold
    assert plan(8192, 8192) == [0]
    # ragged size: steps stop when the next panel would reach the end
    assert plan(5197, 2048) == [0, 128, 640, 1664, 2688, 3712]
    assert plan(5197, 0) == [0, 128, 640, 1664, 2688, 3712, 4736]  # the last 461 columns are the tail
    # wider later steps: first step 128, second 512, then 2048 while 2 * 2048 rows remain behind them, 1024 again at the end
    assert plan(16384, 4096, 2048) == [0, 128, 640, 2688, 4736, 6784, 8832, 10880, 11904, 12928]
    for n in (2049, 3000, 10240, 16384, 20000):
        for tail in (0, 1024, 4096, 1 << 30):
            j = plan(n, tail)
            assert j[0] == 0 and all(b > a for a, b in zip(j, j[1:])) and j[-1] < n
            assert all((b - a) % 128 == 0 for a, b in zip(j, j[1:]))

    # LU leaf: 64 columns while ceil(rows / 512) workgroups are resident (fp64), narrower / taller shapes after that
    w = lambda rows, dt, cap: lib.faer_hip_debug_lu_leaf_width(C.c_size_t(rows), C.c_int(dt), cap)  # noqa: E731
    assert w(16384, F.DTYPE_F64, 32) == 64 and w(16384, F.DTYPE_F64, 256) == 64
    assert w(16385, F.DTYPE_F64, 32) == 32  # 33 workgroups of 512 rows do not fit 32 CUs: 1024 rows per workgroup
    assert w(200000, F.DTYPE_F64, 256) == 32 and w(600000, F.DTYPE_F64, 256) == 8
    assert w(1048576, F.DTYPE_F64, 256) == 8 and w(1048577, F.DTYPE_F64, 256) == 0
    assert w(300000, F.DTYPE_F32, 256) == 32 and w(2097152, F.DTYPE_F32, 256) == 8 and w(2097153, F.DTYPE_F32, 256) == 0
    # distributed LU: the look-ahead panel goes to the 32-CU panel stream only while its leaves keep their whole-chip width
    # there (beyond 131072 fp64 rows they would not fit at all and the leaf would abort: ADVICE r02)
    ok = lambda rows, dt: lib.faer_hip_debug_dist_two_streams_ok(C.c_size_t(rows), C.c_int(dt), 32, 256)  # noqa: E731
    assert ok(16384, F.DTYPE_F64) == 1 and ok(16385, F.DTYPE_F64) == 0 and ok(131072, F.DTYPE_F64) == 0
    assert ok(200000, F.DTYPE_F64) == 0 and ok(2000000, F.DTYPE_F64) == 0 and ok(0, F.DTYPE_F64) == 1
    assert ok(32768, F.DTYPE_F32) == 1 and ok(32769, F.DTYPE_F32) == 0


def test_header_is_valid_c99_and_cxx17(tmp_path):
    """include/faer_hip.h is the drop-in boundary: it must compile on its own as plain C and as C++"""
    import shutil
    import subprocess

    for comp, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++17", "cpp")):
        if shutil.which(comp) is None:
            pytest.skip(f"{comp} not available")
        src = tmp_path / f"hdr.{ext}"
        src.write_text('#include "faer_hip.h"\nint main(void) { return 0; }\n')
        subprocess.check_call([comp, std, "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
