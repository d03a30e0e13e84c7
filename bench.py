#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native dense backend for faer.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gemm|llt|lu|qr|gemv|fplu|cpqr|tridiag|bidiag|hess] [--no-extras] [--no-cpu]

Metric (BASELINE.json): achieved fp64 GFLOP/s.  A "step" is one pass of the hot path over one batch of
synthetic input that is already resident in HBM when the timed region starts:

  gemm (default; BASELINE.json configs[1]) : C = A * B, fp64, N = 8192, column major, Accum::Replace
  llt  (configs[2])                         : in-place lower Cholesky of a 16384^2 SPD matrix (restored from a
                                              pristine copy before every step; the copy is timed separately and
                                              subtracted)
  lu   (configs[3], one GPU)                : in-place partial-pivot LU of a 16384^2 matrix
  qr   (configs[4])                         : fp32 Householder QR of a 1e6 x 256 matrix

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the gemm workload is sharded by
block columns of C with no data-path collective: every rank multiplies the same A by
its own N x 8192 slice of B => weak scaling; `value` is the whole-job rate (sum of all ranks' flops over the
slowest rank's time).  Rank 0 prints ONE JSON line.

The line also carries
  "roofline":     the dominant kernel (the MFMA GEMM) against the fp64 MFMA peak, timed with HIP events on the
                  stream the kernel runs on;
  "cpu_baseline": the CPU oracle (a port of faer's algorithm, see oracle/) timed on a bounded sample on this
                  host -- a reported baseline, NOT the thing measured and not faer itself (no Rust toolchain);
  "others":       one-shot rates of the other hot-path workloads on this GPU (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime maps streams onto 4 hardware queues by default.  A rank of the distributed LU / Cholesky has the caller's
# stream, the library's two CU-masked look-ahead streams, the transport's stream and torch.distributed's own: streams that
# share a queue serialise (profiles/r03_qr_stream_order.txt: whatever was created fifth ran 1.3-6x slower).  Read by the
# runtime at its first call, i.e. after this line; measured neutral on one GPU (LLT 39.06 / 38.74, LU 123.6 / 123.2, QR
# 2.016 / 2.015 ms with 4 / 8).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# The factorizations are chains of launches.  Which kernel dominates, its share of the library's kernel time, its average
# launch duration and its launch count are PARSED at run time from the committed rocprofv3 kernel trace of
# `bench.py --workload X --steps 10 --warmup 2` (profiles/rNN_X_kernel_stats.csv, newest round first) -- they describe that
# recorded run, not this one, and the JSON says so (`source`, `not_measured_this_run`).
PROFILE_ROUNDS = ("r04", "r03", "r02")
BOUND_OF = {  # what bounds the dominant kernel of each chain
    "gemm_kernel": "mfma", "getrf_panel": "latency", "getrf_wpanel": "latency", "qr_panel": "latency", "tq_update": "hbm", "tq_gram": "mfma", "tq_panel": "latency",
    "trsm_leaf": "latency", "potrf_leaf": "latency",
}


def dominant_from_profile(workload):
    import csv

    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{workload}_kernel_stats.csv")
        if not os.path.exists(path):
            continue
        try:
            rows = [r for r in csv.DictReader(open(path)) if "fh::" in r["Name"]]
            total = sum(float(r["TotalDurationNs"]) for r in rows)
            top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
            name = top["Name"].split("(")[0].replace("void ", "")
            bound = next((b for k, b in BOUND_OF.items() if k in name), "latency")
            return {"dominant_kernel": name, "dominant_kernel_bound": bound,
                    "dominant_kernel_share_of_device_time": round(float(top["TotalDurationNs"]) / total, 4),
                    "dominant_kernel_launch_ms": round(float(top["AverageNs"]) * 1e-6, 4), "dominant_kernel_calls_in_trace": int(top["Calls"]),
                    "source": f"profiles/{rnd}_{workload}_kernel_stats.csv", "not_measured_this_run": True}
        except Exception:
            continue
    return None


FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (AMD datasheet), dense
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="gemm", choices=["gemm", "llt", "lu", "qr", "gemv", "fplu", "cpqr", "tridiag", "bidiag", "hess"])
    ap.add_argument("--n", type=int, default=0, help="override the matrix size (testing only)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pause", type=float, default=0.0, help="seconds of idle time before each entry of `others` (diagnostic)")
    ap.add_argument("--transport", default="torch", choices=["torch", "rccl"],
                    help="multi-GPU lu/llt: broadcast through torch.distributed (RCCL under the nccl backend) or the library's own "
                         "RCCL transport (ncclBroadcast on a dedicated stream, no Python callback in the loop)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import __graft_entry__ as ge

    F = ge.load_package()
    L = F.lib()  # fails loudly if libfaer_hip.so is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1:
        assert world == args.gpus, f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})"
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    torch.cuda.set_device(local_rank)
    F.use_torch_stream()
    dev = torch.device("cuda", local_rank)

    rccl = None
    if dist is not None and args.transport == "rccl" and args.workload in ("lu", "llt"):
        # the 128-byte ncclUniqueId of the library's own communicator travels through the torch process group
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(F.RcclTransport.unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        rccl = F.RcclTransport(bytes(idt.cpu().numpy().tobytes()), rank, world)

    def barrier():
        if dist is not None:
            dist.barrier()
        F.synchronize()
        torch.cuda.synchronize()

    def colmajor(m, n, dtype, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        return torch.randn((n, m), dtype=dtype, device=dev, generator=g).t()

    # ------------------------------------------------------------------ workloads
    def make_workload(name, n_override=0):
        """returns (step_fn, flops_per_step, overhead_fn, label, dtype_name)"""
        if name == "gemm":
            n = n_override or 8192
            a = colmajor(n, n, torch.float64, 1)               # replicated on every rank
            b = colmajor(n, n, torch.float64, 2 + rank)         # this rank's block columns of B
            c = torch.empty((n, n), dtype=torch.float64, device=dev).t()

            def step():
                F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0)

            return step, 2.0 * n ** 3, None, f"dgemm_f64_n{n}", "f64"
        if name == "gemv":
            # level-2 shape of the same entry point (matmul with one rhs column): an HBM stream, not an MFMA kernel
            n = n_override or 16384
            a = colmajor(n, n, torch.float64, 6)
            xv = colmajor(n, 1, torch.float64, 7)
            yv = torch.empty((1, n), dtype=torch.float64, device=dev).t()

            def step():
                F.matmul(yv, F.ACCUM_REPLACE, a, xv, 1.0)

            return step, 2.0 * n * n, None, f"dgemv_f64_n{n}", "f64"
        if name == "llt" and world > 1:
            # 1-D block-cyclic columns over the ranks, one RCCL broadcast per factored column panel, look-ahead
            # (csrc/dist_llt.h); the total work is fixed => strong scaling.  The SPD matrix is built per rank from
            # the same generator state: rank r keeps its own block columns of G G^T + n I.
            n = n_override or 16384
            nb = 512  # (1024 through round 3: with the panel travelling in row chunks the narrower step costs nothing on the chain and halves the panel per step)
            gmat = colmajor(n, n, torch.float64, 3)
            cols = torch.cat([torch.arange(b * nb, min(n, (b + 1) * nb), device=dev) for b in range(rank, (n + nb - 1) // nb, world)])
            a = (gmat @ gmat[cols].t())
            a[cols, torch.arange(len(cols), device=dev)] += n
            a = a.t().contiguous().t()
            del gmat
            work = a.clone()
            L.faer_hip_dist_llt_ws_scalars.restype = C.c_size_t
            ws = torch.empty(L.faer_hip_dist_llt_ws_scalars(C.c_size_t(n), C.c_size_t(nb), C.c_int(F.DTYPE_F64)), dtype=torch.float64, device=dev)

            def step():
                work.copy_(a)
                if rccl is not None:
                    F.dist_llt(work, n, nb, rank, world, panel_ws=ws, transport=rccl)
                else:
                    F.dist_llt(work, n, nb, rank, world, lambda t, root: dist.broadcast(t, src=root), panel_ws=ws,
                               ibcast=lambda t, root: dist.broadcast(t, src=root, async_op=True))

            return step, n ** 3 / 3.0 / world, lambda: work.copy_(a), f"llt_f64_n{n}_blockcyclic{nb}", "f64"
        if name == "llt":
            n = n_override or 16384
            a = colmajor(n, n, torch.float64, 3)
            spd = (a @ a.t() + n * torch.eye(n, dtype=torch.float64, device=dev)).t()  # bench.rs:1511-1513
            del a
            work = spd.clone()

            def step():
                work.copy_(spd)
                F.llt_factor_in_place(work)

            return step, n ** 3 / 3.0, lambda: work.copy_(spd), f"llt_f64_n{n}", "f64"
        if name == "lu":
            n = n_override or 16384
            if world > 1:
                # BASELINE.json configs[3]: 1-D block-cyclic columns over the ranks, RCCL broadcast of each factored
                # panel (csrc/dist_lu.h); the total work is fixed => strong scaling
                nb = 512
                ncols = F.dist_local_ncols(n, nb, rank, world)
                a = colmajor(n, ncols, torch.float64, 40 + rank)
                work = a.clone()
                nsc = L.faer_hip_dist_panel_ws_scalars(C.c_size_t(n), C.c_size_t(nb), C.c_int(F.DTYPE_F64))
                ws = torch.empty(nsc, dtype=torch.float64, device=dev)

                def step():
                    work.copy_(a)
                    if rccl is not None:
                        F.dist_partial_piv_lu(work, n, nb, rank, world, panel_ws=ws, transport=rccl)
                    else:
                        F.dist_partial_piv_lu(work, n, nb, rank, world, lambda t, root: dist.broadcast(t, src=root), panel_ws=ws,
                                              ibcast=lambda t, root: dist.broadcast(t, src=root, async_op=True))

                return step, 2.0 * n ** 3 / 3.0 / world, lambda: work.copy_(a), f"lu_f64_n{n}_blockcyclic{nb}", "f64"
            a = colmajor(n, n, torch.float64, 4)
            work = a.clone()

            def step():
                work.copy_(a)
                F.partial_piv_lu_factor_in_place(work)

            return step, 2.0 * n ** 3 / 3.0, lambda: work.copy_(a), f"lu_f64_n{n}", "f64"
        if name in ("qr", "qrmax"):
            # BASELINE config Q says 1e6 x 256, but faer's own rank test rejects every fp32 column once
            # 16 * eps * nrows >= 1 (nrows >= 524288; qr/no_pivoting/factor.rs:52-58), i.e. the reference does no
            # factorization there (rank 0, reproduced by our library and covered by tests).  The rate is therefore
            # quoted on the largest round tall-skinny shape the reference really factors.
            # ("qrmax": 524287 rows, the LARGEST height the reference's rank test still accepts in fp32)
            m, n = (n_override or (524287 if name == "qrmax" else 500000)), 256
            a = colmajor(m, n, torch.float32, 5)
            work = a.clone()
            bs = F.qr_recommended_block_size(m, n, np.float32)
            h = torch.zeros((min(m, n), bs), dtype=torch.float32, device=dev).t()

            def step():
                work.copy_(a)
                F.qr_factor_in_place(work, h)

            return step, 2.0 * m * n * n - 2.0 / 3.0 * n ** 3, lambda: work.copy_(a), f"qr_f32_{m}x{n}", "f32"
        if name == "fplu":
            # LU with full pivoting, a level-2 (HBM bound) algorithm; algorithmic bytes =
            # one read + one write of the trailing matrix per step
            n = n_override or 4096
            a = colmajor(n, n, torch.float64, 6)
            work = a.clone()

            def step():
                work.copy_(a)
                F.full_piv_lu_factor_in_place(work)

            return step, 2.0 * n ** 3 / 3.0, lambda: work.copy_(a), f"full_piv_lu_f64_n{n}", "f64"
        if name == "cpqr":
            # QR with column pivoting, HBM bound like the full-pivot LU
            n = n_override or 4096
            a = colmajor(n, n, torch.float64, 7)
            work = a.clone()
            bs = F.qr_recommended_block_size(n, n, np.float64)
            h = torch.zeros((n, bs), dtype=torch.float64, device=dev).t()

            def step():
                work.copy_(a)
                F.colpiv_qr_factor_in_place(work, h)

            return step, 4.0 * n ** 3 / 3.0, lambda: work.copy_(a), f"colpiv_qr_f64_n{n}", "f64"
        if name == "tridiag":
            # tridiagonalization (evd/tridiag.rs:274), HBM bound: per column the remaining
            # lower triangle is read and written once (4 n^3 / 3 flop in total)
            n = n_override or 4096
            a = colmajor(n, n, torch.float64, 8)
            a = (a + a.t()).t().contiguous().t()
            work = a.clone()
            h = torch.zeros((n - 1, 32), dtype=torch.float64, device=dev).t()

            def step():
                work.copy_(a)
                F.tridiag_in_place(work, h)

            return step, 4.0 * n ** 3 / 3.0, lambda: work.copy_(a), f"tridiag_f64_n{n}", "f64"
        if name == "bidiag":
            # bidiagonalization (svd/bidiag.rs:47), HBM bound: per column the trailing matrix is
            # read + written once and read once more (8 n^3 / 3 flop in total for a square matrix)
            n = n_override or 4096
            a = colmajor(n, n, torch.float64, 9)
            work = a.clone()
            hl = torch.zeros((n, 32), dtype=torch.float64, device=dev).t()
            hr = torch.zeros((n - 1, 32), dtype=torch.float64, device=dev).t()

            def step():
                work.copy_(a)
                F.bidiag_in_place(work, hl, hr)

            return step, 8.0 * n ** 3 / 3.0, lambda: work.copy_(a), f"bidiag_f64_n{n}", "f64"
        if name == "hess":
            # Hessenberg reduction (evd/hessenberg.rs:549), HBM bound: per column A22 is read +
            # written + read and the rows above are read twice + written (10 n^3 / 3 flop in total)
            n = n_override or 4096
            a = colmajor(n, n, torch.float64, 10)
            work = a.clone()
            h = torch.zeros((n - 1, 32), dtype=torch.float64, device=dev).t()

            def step():
                work.copy_(a)
                F.hessenberg_in_place(work, h)

            return step, 10.0 * n ** 3 / 3.0, lambda: work.copy_(a), f"hessenberg_f64_n{n}", "f64"
        raise ValueError(name)

    def timed(fn, steps, warmup):
        """K steps bracketed by barrier + synchronize; returns (seconds, seconds from HIP events)"""
        for _ in range(warmup):
            fn()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        barrier()
        dt = time.perf_counter() - t0
        return dt, e0.elapsed_time(e1) * 1e-3

    step, flops, overhead, label, dtype_name = make_workload(args.workload, args.n)
    dt, dt_ev = timed(step, args.steps, args.warmup)
    if overhead is not None:  # restoring the input is not part of the factorization
        odt, odt_ev = timed(overhead, args.steps, 1)
        dt, dt_ev = max(dt - odt, 1e-9), max(dt_ev - odt_ev, 1e-9)
    per_rank = None
    if dist is not None:
        t = torch.tensor([dt, dt_ev], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_ev = t[0].item(), t[1].item()
        if args.workload in ("lu", "llt"):
            # per rank, from the LAST timed factorization: device time of the call, of the panels the rank owned (the serial
            # chain of a 1-D block-cyclic factorization) and -- with the built-in transport -- inside ncclBroadcast, plus the
            # number of ranks RCCL itself reports: what a scaling curve is read against
            ds = F.dist_last_stats()
            rs = rccl.stats() if rccl is not None else {"ncclCommCount": -1, "broadcasts": 0, "bytes": 0.0, "bcast_device_ms": 0.0}
            mine = torch.tensor([ds["total_device_ms"], ds["panel_device_ms"], ds["panels_owned"], rs["ncclCommCount"], rs["broadcasts"],
                                 rs["bytes"], rs["bcast_device_ms"] / max(args.steps + args.warmup, 1)], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            allr = torch.stack(allr).cpu().numpy()
            per_rank = {"total_device_ms": [round(float(x), 3) for x in allr[:, 0]],
                        "panel_device_ms": [round(float(x), 3) for x in allr[:, 1]],
                        "panels_owned": [int(x) for x in allr[:, 2]],
                        "update_and_wait_device_ms": [round(float(a - b), 3) for a, b in zip(allr[:, 0], allr[:, 1])],
                        "ncclCommCount": [int(x) for x in allr[:, 3]],
                        "bcast_device_ms_per_factorization": [round(float(x), 3) for x in allr[:, 6]],
                        "transport": args.transport}
    ms_per_step = dt / args.steps * 1e3
    value = flops * world * args.steps / dt / 1e9  # whole-job GFLOP/s

    out = {
        "metric": "achieved fp64 GFLOP/s (and % MFMA peak), GEMM + LU/Cholesky, N=16384",
        "value": round(value, 1),
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if (args.workload in ("lu", "llt") and world > 1) else "weak",
        "vs_baseline": None,
        "dtype": dtype_name,
        "data": "synthetic",
        "config": {"workload": label, "layout": "column-major, resident in HBM",
                   "sharding": "none" if world == 1 else (
                       f"1-D block-cyclic columns over {world} GPUs, one RCCL broadcast per factored panel, look-ahead" if args.workload in ("lu", "llt")
                       else f"block columns of C over {world} GPUs, no collective")},
    }

    if per_rank is not None:
        out["per_rank"] = per_rank
    if rank == 0:
        # ---------------------------------------------------------------- roofline of the dominant kernel
        if args.workload == "gemm":
            launch_s = dt_ev / args.steps  # one step == one launch of the MFMA GEMM kernel
            achieved = flops / launch_s / 1e12
            # HBM-side traffic needs the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
            # tools/gpu_pmc.sh): it cannot be measured inside this process.  The figure below is READ from the last
            # committed PMC run of this same kernel and labelled with its source; it is not a measurement of this run.
            pmc, pmc_src = None, None
            pmc_path = os.path.join(ROOT, "profiles", "pmc_gemm_latest.json")
            if os.path.exists(pmc_path):
                try:
                    pj = json.load(open(pmc_path))
                    pmc = pj.get("hbm_bytes_per_launch")
                    pmc_src = f"profiles/pmc_gemm_latest.json ({pj.get('recorded', 'round 1')}; rocprofv3 --pmc, not this run)"
                except Exception:
                    pmc = None
            # the kernel's name (tile shape) from the committed kernel trace of this same command, like the other workloads
            kname = (dominant_from_profile("gemm") or {}).get("dominant_kernel", "fh::gemm_kernel_p<double, 128, 256, 16, 2, 4, false, true, 1>")
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": round(achieved, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": pmc, "traffic_source": pmc_src,
                               "algorithmic_flops_per_launch": flops, "launch_ms": round(launch_s * 1e3, 4)}
        elif args.workload == "qr":
            # HBM bound as specified: algorithmic bytes = the matrix read and written once
            gbs = 2.0 * 500000 * 256 * 4 / (dt / args.steps) / 1e9 if not args.n else None
            out["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1) if gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(gbs / HBM_PEAK_GBS, 4) if gbs else None, "traffic": None}
            out["roofline"].update(dominant_from_profile("qr") or {})
        else:
            # the factorizations are chains of kernels: the whole-chain rate against the matrix-core peak of their trailing
            # updates; the dominant-kernel fields come from the committed kernel trace
            peak = FP64_MFMA_PEAK_TFLOPS if dtype_name == "f64" else FP32_MFMA_PEAK_TFLOPS
            out["roofline"] = {"bound": "mfma", "achieved": round(value / 1e3, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(value / 1e3 / peak, 4), "traffic": None}
            out["roofline"].update(dominant_from_profile(args.workload) or {})
            if not args.no_extras:
                n = 8192
                a, b = colmajor(n, n, torch.float64, 11), colmajor(n, n, torch.float64, 12)
                c = torch.empty((n, n), dtype=torch.float64, device=dev).t()
                ms = L.faer_hip_time_gemm_ms(C.c_int(F.DTYPE_F64), C.c_size_t(n), C.c_size_t(n), C.c_size_t(n),
                                             C.c_void_p(c.data_ptr()), C.c_ssize_t(n), C.c_void_p(a.data_ptr()), C.c_ssize_t(n),
                                             C.c_void_p(b.data_ptr()), C.c_ssize_t(n), 5)
                out["dgemm_n8192_sustained_TFLOP/s"] = round(2.0 * n ** 3 / (ms * 1e-3) / 1e12, 2)
                del a, b, c

        # ---------------------------------------------------------------- other hot-path workloads (one GPU)
        if world == 1 and not args.no_extras:
            others = {}
            del step
            torch.cuda.empty_cache()
            only = os.environ.get("BENCH_OTHERS")  # diagnostic: a comma-separated subset
            for name in ("gemm", "llt", "lu", "qr", "gemv", "fplu", "cpqr", "tridiag", "bidiag", "hess", "qrmax"):
                if name == args.workload or (only and name not in only.split(",")):
                    continue
                try:
                    st, fl, ov, lb, dn = make_workload(name)
                    if args.pause > 0:
                        # the chip leaves a long MFMA-bound run (the headline, LLT, LU) at reduced clocks for a while and the
                        # latency-bound workloads after it measured up to 40 % slower than on their own: every entry of
                        # `others` starts from an idle chip (documented in DESIGN.md 6c; --pause 0 restores back-to-back runs)
                        torch.cuda.synchronize()
                        time.sleep(args.pause)
                    reps = 5
                    t, _ = timed(st, reps, 2)
                    if ov is not None:
                        t = max(t - timed(ov, reps, 1)[0], 1e-9)
                    t = t * 3 / reps  # (the formulas below are written per 3 repetitions)
                    rate = fl * 3 / t / 1e9
                    peak = FP64_MFMA_PEAK_TFLOPS if dn == "f64" else FP32_MFMA_PEAK_TFLOPS
                    others[lb] = {"GFLOP/s": round(rate, 1), "ms": round(t / 3 * 1e3, 3),
                                  "frac_of_mfma_peak": round(rate / 1e3 / peak, 4), "reps": reps}
                    if name in ("llt", "lu"):
                        # per-workload roofline object: `achieved` is the whole-factorization rate (measured now) against the
                        # fp64 matrix-core peak that bounds its trailing updates; the dominant-kernel fields are read from the
                        # committed kernel trace named in `source`
                        others[lb]["roofline"] = {"bound": "mfma", "achieved": round(rate / 1e3, 2), "peak": peak, "unit": "TFLOP/s",
                                                  "frac": round(rate / 1e3 / peak, 4)}
                        others[lb]["roofline"].update(dominant_from_profile(name) or {})
                    if args.workload == "gemm" and name in ("llt", "lu"):
                        others[lb]["frac_of_dgemm_sustained"] = round(rate / value, 4)  # BASELINE target: >= 0.6
                        others[lb]["roofline"]["frac_of_dgemm_sustained"] = others[lb]["frac_of_dgemm_sustained"]
                    if name == "gemv":  # HBM bound: algorithmic bytes = the matrix, read once
                        gbs = (fl / 2.0) * 8 * 3 / t / 1e9
                        others[lb] = {"GB/s": round(gbs, 1), "ms": round(t / 3 * 1e3, 3), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
                    if name == "cpqr":  # HBM bound: the trailing matrix is read and written once per step
                        nn = int(round((fl * {"cpqr": 0.75, "fplu": 1.5, "tridiag": 0.75, "bidiag": 0.375, "hess": 0.3}[name]) ** (1.0 / 3.0)))  # the n of make_workload
                        gbs = sum(2.0 * (nn - k) ** 2 * 8 for k in range(nn)) * 3 / t / 1e9
                        others[lb] = {"GFLOP/s": round(rate, 1), "ms": round(t / 3 * 1e3, 3), "GB/s_algorithmic": round(gbs, 1),
                                      "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
                    if name == "fplu":  # HBM bound: the trailing matrix is read and written once per step
                        nn = int(round((fl * {"cpqr": 0.75, "fplu": 1.5, "tridiag": 0.75, "bidiag": 0.375, "hess": 0.3}[name]) ** (1.0 / 3.0)))  # the n of make_workload
                        gbs = sum(2.0 * (nn - k) ** 2 * 8 for k in range(1, nn)) * 3 / t / 1e9
                        others[lb] = {"GFLOP/s": round(rate, 1), "ms": round(t / 3 * 1e3, 3), "GB/s_algorithmic": round(gbs, 1),
                                      "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
                    if name == "tridiag":  # HBM bound: the remaining lower triangle is read and written once per column
                        nn = int(round((fl * {"cpqr": 0.75, "fplu": 1.5, "tridiag": 0.75, "bidiag": 0.375, "hess": 0.3}[name]) ** (1.0 / 3.0)))  # the n of make_workload
                        gbs = sum((nn - k - 2) ** 2 * 8.0 for k in range(nn - 2)) * 3 / t / 1e9
                        others[lb] = {"GFLOP/s": round(rate, 1), "ms": round(t / 3 * 1e3, 3), "GB/s_algorithmic": round(gbs, 1),
                                      "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
                    if name == "bidiag":  # HBM bound: the trailing matrix is read + written once and read once more per column
                        nn = int(round((fl * {"cpqr": 0.75, "fplu": 1.5, "tridiag": 0.75, "bidiag": 0.375, "hess": 0.3}[name]) ** (1.0 / 3.0)))  # the n of make_workload
                        gbs = sum(3.0 * (nn - k - 1) ** 2 * 8.0 for k in range(nn)) * 3 / t / 1e9
                        others[lb] = {"GFLOP/s": round(rate, 1), "ms": round(t / 3 * 1e3, 3), "GB/s_algorithmic": round(gbs, 1),
                                      "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
                    if name == "hess":  # HBM bound: A22 read + written + read, the k + 1 rows above read twice + written, per column
                        nn = int(round((fl * {"cpqr": 0.75, "fplu": 1.5, "tridiag": 0.75, "bidiag": 0.375, "hess": 0.3}[name]) ** (1.0 / 3.0)))  # the n of make_workload
                        gbs = sum(3.0 * ((nn - k - 1) ** 2 + (k + 1) * (nn - k - 1)) * 8.0 for k in range(nn)) * 3 / t / 1e9
                        others[lb] = {"GFLOP/s": round(rate, 1), "ms": round(t / 3 * 1e3, 3), "GB/s_algorithmic": round(gbs, 1),
                                      "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
                    if name in ("qr", "qrmax"):  # HBM bound as specified: algorithmic bytes 2 m n sizeof(f32)
                        gbs = 2.0 * (524287 if name == "qrmax" else 500000) * 256 * 4 * 3 / t / 1e9
                        others[lb]["GB/s_algorithmic"] = round(gbs, 1)
                        if name == "qr":
                            others[lb]["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                      "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                      "algorithmic_bytes": 2.0 * 500000 * 256 * 4}
                            others[lb]["roofline"].update(dominant_from_profile("qr") or {})
                    del st, ov
                    if not os.environ.get("BENCH_KEEP_CACHE"):  # diagnostic
                        torch.cuda.empty_cache()
                except Exception as ex:  # keep the headline line even if an extra fails
                    others[name] = {"error": str(ex)[:200]}
            out["others"] = others

        # ---------------------------------------------------------------- CPU baseline (oracle port, bounded sample)
        if world == 1 and not args.no_cpu:
            from oracle import oracle as orc

            ncpu = os.cpu_count() or 1
            nthr = min(ncpu, 64)  # the port's column-parallel loops stop scaling well before 256 threads
            os.environ["OMP_NUM_THREADS"] = str(nthr)
            rng = np.random.default_rng(0)
            hw = np.zeros((256, 256), order="F")
            ha = np.asfortranarray(rng.standard_normal((256, 256)))
            orc.matmul(hw, ha, ha)  # warm (thread pool start-up)
            # bounded sample of the SAME workloads (about 10-30 s of CPU work in total)
            n_mm, n_llt = 4096, 3072
            ha, hb = np.asfortranarray(rng.standard_normal((n_mm, n_mm))), np.asfortranarray(rng.standard_normal((n_mm, n_mm)))
            hc = np.zeros((n_mm, n_mm), order="F")
            t0 = time.perf_counter()
            orc.matmul(hc, ha, hb)
            t_mm = time.perf_counter() - t0
            hs = ha[:n_llt, :n_llt]
            spd = np.asfortranarray(hs @ hs.T + n_llt * np.eye(n_llt))
            t0 = time.perf_counter()
            orc.llt_in_place(spd)
            t_llt = time.perf_counter() - t0
            # second proxy (SURVEY.md section 8d ii): the host's optimised BLAS / LAPACK
            # (scipy -> OpenBLAS) on the same shapes -- what a tuned CPU library reaches on these cores; still not faer
            blas = {}
            try:
                import scipy.linalg.blas as sblas
                import scipy.linalg.lapack as slap

                def best(fn, reps=3):
                    fn()
                    tb = 1e30
                    for _ in range(reps):
                        t0 = time.perf_counter()
                        fn()
                        tb = min(tb, time.perf_counter() - t0)
                    return tb

                nb_ = 4096
                xa = np.asfortranarray(rng.standard_normal((nb_, nb_)))
                xs = np.asfortranarray(xa @ xa.T + nb_ * np.eye(nb_))
                blas["dgemm_n4096_GFLOP/s"] = round(2.0 * nb_ ** 3 / best(lambda: sblas.dgemm(1.0, xa, xa)) / 1e9, 1)
                blas["dpotrf_n4096_GFLOP/s"] = round(nb_ ** 3 / 3.0 / best(lambda: slap.dpotrf(xs, lower=1)) / 1e9, 1)
                blas["dgetrf_n4096_GFLOP/s"] = round(2.0 * nb_ ** 3 / 3.0 / best(lambda: slap.dgetrf(xa)) / 1e9, 1)
                xq = np.asfortranarray(rng.standard_normal((100000, 256)).astype(np.float32))
                blas["sgeqrf_100000x256_GFLOP/s"] = round((2.0 * 100000 * 256 ** 2 - 2.0 / 3.0 * 256 ** 3) / best(lambda: slap.sgeqrf(xq), 2) / 1e9, 1)
                try:
                    from threadpoolctl import threadpool_info

                    blas["threads"] = max((d.get("num_threads", 0) for d in threadpool_info()), default=0)
                except Exception:
                    blas["threads"] = None
                blas["kind"] = "proxy: scipy/OpenBLAS on the same host (not faer)"
            except Exception as ex:
                blas = {"error": str(ex)[:200]}
            out["cpu_baseline_openblas"] = blas
            out["cpu_baseline"] = {"value": round(2.0 * n_mm ** 3 / t_mm / 1e9, 2), "unit": "GFLOP/s", "cores": nthr,
                                   "kind": "port",
                                   "sample": f"oracle (C restatement of faer's algorithm, OpenMP over columns, {nthr} threads of "
                                             f"{ncpu} host cores) fp64 matmul n={n_mm}, 1 rep ({t_mm:.2f} s); "
                                             f"llt n={n_llt}: {n_llt ** 3 / 3.0 / t_llt / 1e9:.2f} GFLOP/s ({t_llt:.2f} s)",
                                   "note": "faer itself cannot be built here (no Rust toolchain); proxy, see BASELINE.md"}
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
