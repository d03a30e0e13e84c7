"""One-shot GPU diagnostics: MFMA issue-rate ceiling, GEMM timings per tile variant, factorization timings.
Writes a plain-text table to stdout (redirect into gpurun_out/)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
which = sys.argv[1:] or ["peak", "gemm", "llt", "lu"]


def colmajor(m, n, dtype=torch.float64, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn((n, m), dtype=dtype, device="cuda", generator=g).t()


if "peak" in which:
    for name, dt in (("f64", F.DTYPE_F64), ("f32", F.DTYPE_F32)):
        L.faer_hip_mfma_peak_tflops(C.c_int(dt), 2000)
        print(f"mfma_peak {name}: {L.faer_hip_mfma_peak_tflops(C.c_int(dt), 20000):.1f} TFLOP/s", flush=True)

if "gemm" in which:
    for dtype, dt, name in ((torch.float64, F.DTYPE_F64, "f64"), (torch.float32, F.DTYPE_F32, "f32")):
        for n in (2048, 4096, 8192):
            a, b = colmajor(n, n, dtype, 1), colmajor(n, n, dtype, 2)
            c = torch.empty((n, n), dtype=dtype, device="cuda").t()
            for variant in (1, 2, 11, 12):
                L.faer_hip_set_gemm_variant(variant)
                ms = L.faer_hip_time_gemm_ms(C.c_int(dt), C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), C.c_void_p(c.data_ptr()),
                                             C.c_ssize_t(n), C.c_void_p(a.data_ptr()), C.c_ssize_t(n),
                                             C.c_void_p(b.data_ptr()), C.c_ssize_t(n), 5 if n >= 8192 else 10)
                print(f"gemm {name} N={n} variant={variant}: {ms:.3f} ms  {2.0 * n ** 3 / ms / 1e9:.2f} TFLOP/s", flush=True)
            L.faer_hip_set_gemm_variant(0)
            if dtype == torch.float64:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    torch.matmul(a, b)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 3 * 1e3
                print(f"  (rocBLAS via torch.matmul f64 N={n}: {ms:.3f} ms  {2.0 * n ** 3 / ms / 1e9:.2f} TFLOP/s)", flush=True)
            del a, b, c


def timeit(fn, reps=3):
    fn()
    F.synchronize()
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        F.synchronize()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


if "llt" in which:
    for n in (4096, 8192, 16384):
        a = colmajor(n, n, seed=3)
        spd = (a @ a.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t()
        work = spd.clone()

        def run():
            work.copy_(spd)
            F.llt_factor_in_place(work)

        def copy_only():
            work.copy_(spd)

        t = timeit(run) - timeit(copy_only)
        print(f"llt f64 N={n}: {t * 1e3:.2f} ms  {n ** 3 / 3 / t / 1e12:.2f} TFLOP/s", flush=True)
        del a, spd, work

if "lu" in which:
    for n in (4096, 8192, 16384):
        a = colmajor(n, n, seed=4)
        work = a.clone()

        def run():
            work.copy_(a)
            F.partial_piv_lu_factor_in_place(work)

        def copy_only():
            work.copy_(a)

        t = timeit(run) - timeit(copy_only)
        print(f"lu f64 N={n}: {t * 1e3:.2f} ms  {2 * n ** 3 / 3 / t / 1e12:.2f} TFLOP/s", flush=True)
        del a, work

if "qr" in which:
    m, n = 500000, 256
    a = colmajor(m, n, torch.float32, 5)
    work = a.clone()
    h = torch.zeros((n, 256), dtype=torch.float32, device="cuda").t()

    def run():
        work.copy_(a)
        F.qr_factor_in_place(work, h)

    def copy_only():
        work.copy_(a)

    t = timeit(run) - timeit(copy_only)
    print(f"qr f32 {m}x{n}: {t * 1e3:.2f} ms  {(2.0 * m * n * n - 2.0 / 3 * n ** 3) / t / 1e12:.2f} TFLOP/s", flush=True)
