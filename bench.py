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

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) and no --workload the headline is
BASELINE.json configs[3]: the partial-pivot LU of a 16384^2 matrix, 1-D block-cyclic columns over the N ranks, one
RCCL broadcast per factored panel through the library's own transport (`resolve_run` below) => strong scaling;
`value` is the whole-job rate, `per_rank.ncclCommCount` must equal N, `lu_1gpu_same_run` is the single-GPU LU of
the same size timed on rank 0 in the same run (the N = 1 point of the curve; it is also `others.lu_f64_n16384` of
the --gpus 1 line).  `others` then carries the collective-free weak-scaling GEMM (every rank multiplies the same
A by its own N x 8192 slice of B) and the block-cyclic LU at N = 65536.  Rank 0 prints ONE JSON line.
BENCH_FORCE_DIST=1 with --gpus 1 runs exactly that code path with ONE rank (process group, RCCL communicator, block-cyclic
driver, the N = 65536 entry): a dry run of the multi-GPU default on a one-GPU box (profiles/r05_bench_dryrun_dist_one_rank.json).

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
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02")
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


def resolve_run(gpus, workload, transport):
    """What a command line runs: (workload, transport, defaulted).  One GPU: the DGEMM of BASELINE.json configs[1]; several GPUs
    and no --workload: configs[3], the block-cyclic LU over the library's RCCL transport (a collective-free GEMM would show
    N x with zero RCCL traffic and say nothing about the distributed path -- VERDICT r04 item 4)."""
    defaulted = workload is None
    multi = gpus > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))  # (BENCH_FORCE_DIST=1: the N > 1 code path with ONE rank, a dry run)
    if workload is None:
        workload = "lu" if multi else "gemm"
    if transport is None:
        # (round 6: an explicit --workload lu / llt on several GPUs takes the library's RCCL transport as well -- the torch transport runs a
        # Python callback per broadcast, which measured 201 ms against ~100 for the one-rank Cholesky and is not what a user would deploy)
        transport = "rccl" if (multi and workload in ("lu", "llt")) else "torch"
    return workload, transport, defaulted


def per_rank_dict(allr, transport):
    """`per_rank` of a distributed lu / llt line from the all-gathered rows {total ms, panel ms, panels owned, ncclCommCount,
    broadcasts, bytes, broadcast ms per factorization}: what a scaling curve is read against -- with the library's RCCL
    transport `ncclCommCount` is what RCCL itself reports on every rank and must equal the number of GPUs."""
    return {"total_device_ms": [round(float(r[0]), 3) for r in allr],
            "panel_device_ms": [round(float(r[1]), 3) for r in allr],
            "panels_owned": [int(r[2]) for r in allr],
            "update_and_wait_device_ms": [round(float(r[0] - r[1]), 3) for r in allr],
            "ncclCommCount": [int(r[3]) for r in allr],
            "broadcasts": [int(r[4]) for r in allr],
            "bcast_device_ms_per_factorization": [round(float(r[6]), 3) for r in allr],
            "transport": transport}


FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (AMD datasheet), dense
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["gemm", "sgemm", "llt", "lu", "qr", "qr64", "qrsq", "gemv", "fplu", "cpqr", "tridiag", "bidiag", "hess"],
                    help="default: gemm on one GPU, the block-cyclic lu (RCCL transport) on several (resolve_run)")
    ap.add_argument("--n", type=int, default=0, help="override the matrix size (testing only)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pause", type=float, default=0.0, help="seconds of idle time before each entry of `others` (diagnostic)")
    ap.add_argument("--transport", default=None, choices=["torch", "rccl"],
                    help="multi-GPU lu/llt: broadcast through torch.distributed (RCCL under the nccl backend) or the library's own "
                         "RCCL transport (ncclBroadcast on a dedicated stream, no Python callback in the loop)")
    args = ap.parse_args()
    args.workload, args.transport, defaulted = resolve_run(args.gpus, args.workload, args.transport)

    import numpy as np
    import torch

    import __graft_entry__ as ge

    F = ge.load_package()
    L = F.lib()  # fails loudly if libfaer_hip.so is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = bool(os.environ.get("BENCH_FORCE_DIST")) and args.gpus == 1
    if args.gpus > 1 or force_dist:
        assert world == args.gpus, f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})"
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    torch.cuda.set_device(local_rank)
    F.use_torch_stream()
    dev = torch.device("cuda", local_rank)

    rccl = None
    if dist is not None and args.transport == "rccl" and args.workload in ("lu", "llt"):
        # the 128-byte ncclUniqueId of the library's own communicator travels through the torch process group
        # (a rank that cannot open the library's own communicator -- librccl not loadable, ncclCommInitRank failing -- must not
        # leave the others waiting: every rank reports, and unless ALL succeeded the run falls back to the torch.distributed
        # transport, which is RCCL under the nccl backend as well; the line then says transport = "torch (fallback: ...)")
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        err = ""
        try:
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(F.RcclTransport.unique_id()), dtype=torch.uint8))
        except Exception as ex:
            err = str(ex)[:120]
        dist.broadcast(idt, src=0)
        if not err and int(idt.sum().item()) == 0:
            err = "no ncclUniqueId"
        try:
            if not err:
                rccl = F.RcclTransport(bytes(idt.cpu().numpy().tobytes()), rank, world)
        except Exception as ex:
            err, rccl = str(ex)[:120], None
        okt = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:
            rccl = None
            args.transport = f"torch (fallback: built-in RCCL transport unavailable on some rank{': ' + err if err else ''})"

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
        if name in ("gemm", "gemm4096", "sgemm"):
            # "gemm4096": the mid-size product the factorizations live on; "sgemm": north_star's fp32 MFMA GEMM
            n = n_override or (4096 if name == "gemm4096" else 8192)
            dt_ = torch.float32 if name == "sgemm" else torch.float64
            a = colmajor(n, n, dt_, 1)               # replicated on every rank
            b = colmajor(n, n, dt_, 2 + rank)         # this rank's block columns of B
            c = torch.empty((n, n), dtype=dt_, device=dev).t()

            def step():
                F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0)

            return step, 2.0 * n ** 3, None, (f"sgemm_f32_n{n}" if name == "sgemm" else f"dgemm_f64_n{n}"), ("f32" if name == "sgemm" else "f64")
        if name == "gemv":
            # level-2 shape of the same entry point (matmul with one rhs column): an HBM stream, not an MFMA kernel
            n = n_override or 16384
            a = colmajor(n, n, torch.float64, 6)
            xv = colmajor(n, 1, torch.float64, 7)
            yv = torch.empty((1, n), dtype=torch.float64, device=dev).t()

            def step():
                F.matmul(yv, F.ACCUM_REPLACE, a, xv, 1.0)

            return step, 2.0 * n * n, None, f"dgemv_f64_n{n}", "f64"
        if name == "llt" and dist is not None:
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
            if dist is not None:
                # BASELINE.json configs[3]: 1-D block-cyclic columns over the ranks, RCCL broadcast of each factored
                # panel (csrc/dist_lu.h); the total work is fixed => strong scaling
                nb = int(os.environ.get("BENCH_DIST_NB", "512"))
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
        if name == "qrlit":
            # BASELINE.json configs[4] VERBATIM: 1e6 x 256 fp32.  The reference's own rank test rejects every column from 524288
            # rows on (see "qr" below), so this call does what faer does there -- rank 0, no reflector applied -- and the entry
            # times that, not a factorization (tests/test_gpu_qr.py::test_qr_fp32_reference_rank_test_limit)
            m, n = 1000000, 256
            a = colmajor(m, n, torch.float32, 5)
            work = a.clone()
            bs = F.qr_recommended_block_size(m, n, np.float32)
            h = torch.zeros((min(m, n), bs), dtype=torch.float32, device=dev).t()

            def step():
                work.copy_(a)
                F.qr_factor_in_place(work, h)

            return step, 2.0 * m * n * n - 2.0 / 3.0 * n ** 3, lambda: work.copy_(a), f"qr_f32_{m}x{n}", "f32"
        if name in ("qr", "qrmax"):
            # BASELINE config Q says 1e6 x 256, but faer's own rank test rejects every fp32 column once
            # 16 * eps * nrows >= 1 (nrows >= 524288; qr/no_pivoting/factor.rs:52-58), i.e. the reference does no
            # factorization there (rank 0, reproduced by our library and covered by tests).  The rate is therefore
            # quoted on the largest round tall-skinny shape the reference really factors.
            # ("qrmax": 524287 rows, the LARGEST height the reference's rank test still accepts in fp32)
            m, n = (n_override or (524287 if name == "qrmax" else 500000)), 256
            # faer's Mat pads the column stride to 64 bytes (mat/matown.rs:67-80): 16 floats -- what a faer caller hands over for a height
            # that is not a multiple of 16 (524287); round 6 (rounds 3-5 timed "qrmax" with stride = height: scalar loads in every kernel)
            ldp = (m + 15) // 16 * 16
            g_ = torch.Generator(device=dev).manual_seed(5)
            a = torch.randn((n, ldp), dtype=torch.float32, device=dev, generator=g_)[:, :m].t()
            work = torch.empty((n, ldp), dtype=torch.float32, device=dev)[:, :m].t()
            work.copy_(a)
            bs = F.qr_recommended_block_size(m, n, np.float32)
            h = torch.zeros((min(m, n), bs), dtype=torch.float32, device=dev).t()

            def step():
                work.copy_(a)
                F.qr_factor_in_place(work, h)

            return step, 2.0 * m * n * n - 2.0 / 3.0 * n ** 3, lambda: work.copy_(a), f"qr_f32_{m}x{n}", "f32"
        if name == "qr64":
            # the same tall-skinny shape in the reference's main arithmetic (fp64): the one-pass path's fp64 instantiation (csrc/tsqr.hip,
            # tsqr_factor64).  faer's Mat layout: column stride padded to 64 bytes
            m, n = (n_override or 500000), 256
            ldp = (m + 7) // 8 * 8
            g_ = torch.Generator(device=dev).manual_seed(5)
            a = torch.randn((n, ldp), dtype=torch.float64, device=dev, generator=g_)[:, :m].t()
            work = torch.empty((n, ldp), dtype=torch.float64, device=dev)[:, :m].t()
            work.copy_(a)
            bs = F.qr_recommended_block_size(m, n, np.float64)
            h = torch.zeros((min(m, n), bs), dtype=torch.float64, device=dev).t()

            def step():
                work.copy_(a)
                F.qr_factor_in_place(work, h)

            return step, 2.0 * m * n * n - 2.0 / 3.0 * n ** 3, lambda: work.copy_(a), f"qr_f64_{m}x{n}", "f64"
        if name == "qrsq":
            # square Householder QR (qr/no_pivoting/factor.rs:137-256 on the classic path of csrc/qr.hip, one-pass panels inside the recursion
            # since the end of round 6): not a BASELINE config, kept in the line because it is the QR every square solve runs
            n = n_override or 4096
            a = colmajor(n, n, torch.float64, 7)
            work = a.clone()
            bs = F.qr_recommended_block_size(n, n, np.float64)
            h = torch.zeros((n, bs), dtype=torch.float64, device=dev).t()

            def step():
                work.copy_(a)
                F.qr_factor_in_place(work, h)

            return step, 4.0 / 3.0 * n ** 3, lambda: work.copy_(a), f"qr_f64_n{n}", "f64"
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

    def profiled(fn):
        """one more repetition with the library's kernel-class profile on (HIP events around every launch of the dominant
        classes, on the stream the kernel runs on): {class: {ms, launches, units}} -- measured in THIS run"""
        F.synchronize()
        torch.cuda.synchronize()
        F.prof_begin()
        fn()
        F.synchronize()
        torch.cuda.synchronize()
        return F.prof_end()

    def class_roofline(prof, name, step_ms):
        """roofline object of one factorization from its profiled repetition: the trailing products (or, QR, the update kernel)"""
        if name in ("llt", "lu"):
            c = prof["mfma_products"]
            if c["launches"] == 0 or c["ms"] <= 0:
                return None
            tf = c["units"] / (c["ms"] * 1e-3) / 1e12
            r = {"bound": "mfma", "kernel_class": "gemm_kernel_p (128 x 128 / 128 x 256 tiles): the trailing updates", "achieved": round(tf, 2),
                 "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                 "launches": c["launches"], "launch_ms_avg": round(c["ms"] / c["launches"], 4), "class_ms": round(c["ms"], 3),
                 "class_share_of_step": round(c["ms"] / step_ms, 4), "algorithmic_flops_in_class": c["units"],
                 "measured": "this run: HIP events around every launch of the class, one profiled repetition"}
            side = prof["lu_panel"] if name == "lu" else prof["llt_leaf"]
            if side["launches"]:
                r["chain_kernel"] = {"class": "getrf_wpanel_kernel" if name == "lu" else "potrf_leaf_kernel", "bound": "latency",
                                     "launches": side["launches"], "ms": round(side["ms"], 3),
                                     "us_per_column": round(side["ms"] * 1e3 / max(side["units"], 1.0), 3)}
            return r
        if name == "qr":
            c, g, pk = prof["qr_update"], prof["qr_gram"], prof["qr_panel"]
            if c["launches"] == 0 or c["ms"] <= 0:
                return None
            gbs = c["units"] / (c["ms"] * 1e-3) / 1e9
            r = {"bound": "hbm", "kernel_class": "tq_fused_kernel<1|2> + tq_update_kernel (panel applied, next panel's Gram products formed in the same pass)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "launches": c["launches"], "launch_ms_avg": round(c["ms"] / c["launches"], 4),
                 "class_ms": round(c["ms"], 3), "class_share_of_step": round(c["ms"] / step_ms, 4), "algorithmic_bytes_in_class": c["units"],
                 "measured": "this run: HIP events around every launch of the class, one profiled repetition"}
            if g["launches"] and g["ms"] > 0:
                r["gram_kernel"] = {"launches": g["launches"], "ms": round(g["ms"], 3), "GB/s": round(g["units"] / (g["ms"] * 1e-3) / 1e9, 1)}
            if pk["launches"]:
                r["panel_kernel"] = {"launches": pk["launches"], "ms": round(pk["ms"], 3), "bound": "latency (one workgroup)"}
            return r
        return None

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
            per_rank = per_rank_dict(allr, args.transport)
    ms_per_step = dt / args.steps * 1e3
    value = flops * world * args.steps / dt / 1e9  # whole-job GFLOP/s
    prof_main = None
    if dist is None:
        try:
            prof_main = profiled(step)
        except Exception:
            prof_main = None

    # ---- several GPUs, default run: the collective-free GEMM and the large block-cyclic LU ride along (all ranks take part),
    # and rank 0 times the single-GPU LU of the headline's size: the N = 1 point of the strong-scaling curve, same run
    multi_others, lu_1gpu = None, None
    if dist is not None and defaulted and args.workload == "lu" and not args.no_extras:
        multi_others = {}
        del step
        torch.cuda.empty_cache()
        for name, nn, reps, wu in (("gemm", 0, 10, 2), ("lu", 65536, 2, 1)):
            try:
                st, fl, ov, lb, dn = make_workload(name, nn)
                t, _ = timed(st, reps, wu)
                if ov is not None:
                    t = max(t - timed(ov, reps, 1)[0], 1e-9)
                tt = torch.tensor([t], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t = tt.item()
                rate = fl * world * reps / t / 1e9
                multi_others[lb + ("_block_columns_no_collective" if name == "gemm" else "")] = {
                    "GFLOP/s": round(rate, 1), "ms": round(t / reps * 1e3, 3), "scaling": "weak" if name == "gemm" else "strong", "reps": reps,
                    "frac_of_mfma_peak_per_gpu": round(rate / 1e3 / world / FP64_MFMA_PEAK_TFLOPS, 4)}
                del st, ov
                torch.cuda.empty_cache()
            except Exception as ex:  # (symmetric failures only: every rank runs the same sizes)
                multi_others[name] = {"error": str(ex)[:200]}
        dist.barrier()
    # ---- the N = 1 point of a strong-scaling curve, measured in the SAME run: rank 0 times the single-GPU driver on the same
    # workload (lu / llt) and size while the others wait; `speedup_vs_1gpu_same_run` at top level
    if dist is not None and args.workload in ("lu", "llt"):
        if rank == 0:
            try:
                torch.cuda.empty_cache()
                n1 = args.n or 16384
                if args.workload == "lu":
                    a1 = colmajor(n1, n1, torch.float64, 4)
                    fl1 = 2.0 * n1 ** 3 / 3.0
                else:
                    g1 = colmajor(n1, n1, torch.float64, 3)
                    a1 = (g1 @ g1.t() + n1 * torch.eye(n1, dtype=torch.float64, device=dev)).t()
                    del g1
                    fl1 = n1 ** 3 / 3.0
                w1 = a1.clone()

                def one():
                    w1.copy_(a1)
                    if args.workload == "lu":
                        F.partial_piv_lu_factor_in_place(w1)
                    else:
                        F.llt_factor_in_place(w1)

                def timed_local(fn, reps):
                    fn()
                    F.synchronize(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        fn()
                    F.synchronize(); torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / reps

                t1 = max(timed_local(one, 5) - timed_local(lambda: w1.copy_(a1), 5), 1e-9)
                lu_1gpu = {"workload": f"{args.workload}_f64_n{n1}", "ms": round(t1 * 1e3, 3), "GFLOP/s": round(fl1 / t1 / 1e9, 1),
                           "speedup_of_this_run": round(t1 * 1e3 / ms_per_step, 3)}
                del a1, w1
            except Exception as ex:
                lu_1gpu = {"error": str(ex)[:200]}
        dist.barrier()

    out = {
        "metric": "achieved fp64 GFLOP/s (and % MFMA peak), GEMM + LU/Cholesky, N=16384",
        "value": round(value, 1),
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if (args.workload in ("lu", "llt") and dist is not None) else "weak",
        "vs_baseline": None,
        "dtype": dtype_name,
        "data": "synthetic",
        "config": {"workload": label, "layout": "column-major, resident in HBM",
                   "sharding": "none" if dist is None else (
                       f"1-D block-cyclic columns over {world} GPUs, one RCCL broadcast per factored panel, look-ahead" if args.workload in ("lu", "llt")
                       else f"block columns of C over {world} GPUs, no collective")},
    }

    if per_rank is not None:
        out["per_rank"] = per_rank
    if multi_others is not None:
        out["others"] = multi_others
    if lu_1gpu is not None:
        out[f"{args.workload}_1gpu_same_run"] = lu_1gpu
        if "speedup_of_this_run" in lu_1gpu:
            out["speedup_vs_1gpu_same_run"] = lu_1gpu["speedup_of_this_run"]
    if rank == 0:
        try:
            # idle-chip hand-off between two workgroups.  It separated the pool's two kinds of boxes for most of
            # round 5 and then read 0.39-0.42 us on boxes that ran LU in 104-105 ms and 0.58 us on one that ran it in 87.8: it does
            # NOT classify a box.  What does is in the line already: others.*.roofline.dominant_kernel.chain_kernel.us_per_column,
            # measured in the same run (LU leaf 3.0 against 3.85 us per column, Cholesky leaf 0.475 against 0.56-0.58; profiles/notes/DESIGN_history_r01_r05.md 6e)
            out["xwg_hop_us"] = round(F.xwg_hop_us(2000), 3)
        except Exception:
            out["xwg_hop_us"] = None
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
            if prof_main and prof_main["mfma_products"]["launches"]:
                c0 = prof_main["mfma_products"]
                out["roofline"]["launch_ms_single_profiled_launch"] = round(c0["ms"] / c0["launches"], 4)
        elif args.workload == "qr":
            # HBM bound as specified: algorithmic bytes = the matrix read and written once
            gbs = 2.0 * 500000 * 256 * 4 / (dt / args.steps) / 1e9 if not args.n else None
            out["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1) if gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(gbs / HBM_PEAK_GBS, 4) if gbs else None, "traffic": None,
                               "scope": "whole factorization: algorithmic bytes 2 m n sizeof(f32) over the step time"}
            kr = class_roofline(prof_main, "qr", ms_per_step) if prof_main else None
            if kr:
                out["roofline"]["dominant_kernel"] = kr
            else:
                out["roofline"].update(dominant_from_profile("qr") or {})
        else:
            # the factorizations are chains of kernels: the whole-chain rate against the matrix-core peak of their trailing
            # updates; the dominant-kernel fields come from the committed kernel trace
            peak = FP64_MFMA_PEAK_TFLOPS if dtype_name == "f64" else FP32_MFMA_PEAK_TFLOPS
            out["roofline"] = {"bound": "mfma", "achieved": round(value / 1e3, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(value / 1e3 / peak, 4), "traffic": None,
                               "scope": "whole factorization rate against the matrix-core peak that bounds its trailing updates"}
            kr = class_roofline(prof_main, args.workload, ms_per_step) if prof_main else None
            if kr:
                out["roofline"]["dominant_kernel"] = kr
            else:
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
        if dist is None and not args.no_extras:
            others = {}
            del step
            torch.cuda.empty_cache()
            only = os.environ.get("BENCH_OTHERS")  # diagnostic: a comma-separated subset
            for name in ("gemm", "llt", "lu", "qr", "qr64", "qrsq", "gemm4096", "sgemm", "gemv", "fplu", "cpqr", "tridiag", "bidiag", "hess", "qrmax", "qrlit"):
                if name == args.workload or (only and name not in only.split(",")):
                    continue
                try:
                    st, fl, ov, lb, dn = make_workload(name)
                    if args.pause > 0:
                        # the chip leaves a long MFMA-bound run (the headline, LLT, LU) at reduced clocks for a while and the
                        # latency-bound workloads after it measured up to 40 % slower than on their own: every entry of
                        # `others` starts from an idle chip (documented in profiles/notes/DESIGN_history_r01_r05.md 6c; --pause 0 restores back-to-back runs)
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
                    kr = None
                    if name in ("llt", "lu", "qr", "qr64"):
                        try:
                            kr = class_roofline(profiled(st), "qr" if name == "qr64" else name, t / 3 * 1e3)
                            if kr and name == "qr64":
                                kr["kernel_class"] = "tq_update64_kernel (panel applied: X - P Y and V = P M on the fp64 matrix cores)"
                        except Exception:
                            kr = None
                    if name in ("llt", "lu"):
                        # per-workload roofline object: `achieved` is the whole-factorization rate against the fp64 matrix-core
                        # peak that bounds its trailing updates; `dominant_kernel` is its trailing-product class measured in THIS
                        # run (one more repetition with HIP events around every launch of the class)
                        others[lb]["roofline"] = {"bound": "mfma", "achieved": round(rate / 1e3, 2), "peak": peak, "unit": "TFLOP/s",
                                                  "frac": round(rate / 1e3 / peak, 4)}
                        if kr:
                            others[lb]["roofline"]["dominant_kernel"] = kr
                        else:
                            others[lb]["roofline"].update(dominant_from_profile(name) or {})
                    if name in ("gemm4096", "sgemm"):
                        others[lb]["roofline"] = {"bound": "mfma", "achieved": round(rate / 1e3, 2), "peak": peak, "unit": "TFLOP/s",
                                                  "frac": round(rate / 1e3 / peak, 4), "launch_ms": round(t / 3 * 1e3, 4)}
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
                    if name == "qrlit":
                        others[lb] = {"ms": round(t / 3 * 1e3, 3), "reps": reps,
                                      "note": "BASELINE configs[4] verbatim: the reference's rank test (qr/no_pivoting/factor.rs:52-58) rejects every "
                                              "fp32 column from 524288 rows on, so this is rank 0 / no reflector on either side -- not a "
                                              "factorization rate; see qr_f32_500000x256"}
                    if name == "qr64":  # HBM bound as specified: algorithmic bytes 2 m n sizeof(f64)
                        gbs = 2.0 * 500000 * 256 * 8 * 3 / t / 1e9
                        others[lb]["GB/s_algorithmic"] = round(gbs, 1)
                        others[lb]["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes": 2.0 * 500000 * 256 * 8}
                        if kr:
                            others[lb]["roofline"]["dominant_kernel"] = kr
                    if name in ("qr", "qrmax"):  # HBM bound as specified: algorithmic bytes 2 m n sizeof(f32)
                        gbs = 2.0 * (524287 if name == "qrmax" else 500000) * 256 * 4 * 3 / t / 1e9
                        others[lb]["GB/s_algorithmic"] = round(gbs, 1)
                        if name == "qr":
                            others[lb]["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                      "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                      "algorithmic_bytes": 2.0 * 500000 * 256 * 4,
                                                      # (what else bounds this chain -- DESIGN.md 3.5: the streaming kernels are co-bound by the matrix
                                                      # cores; constants of the algorithm and of the committed PMC pass, not measurements of this run)
                                                      "co_bounds": {"fp32_mfma_flop": 2.0 * 500000 * 256 * 256, "fp64_mfma_flop": 4 * 2.0 * 500000 * 64 * 64 * 10 / 16,
                                                                    "mfma_floor_ms": round(1e3 * (2.0 * 500000 * 256 * 256 / (FP32_MFMA_PEAK_TFLOPS * 1e12)
                                                                                                  + 4 * 2.0 * 500000 * 64 * 64 * 10 / 16 / (FP64_MFMA_PEAK_TFLOPS * 1e12)), 3),
                                                                    "hbm_side_bytes_pmc": 4.24e9, "hbm_floor_ms_at_5TBps": round(4.24e9 / 5e12 * 1e3, 3),
                                                                    "source": "profiles/r06_pmc_qr_summary.json, tools/tall_stream_probe.hip"}}
                            if kr:
                                others[lb]["roofline"]["dominant_kernel"] = kr
                            else:
                                others[lb]["roofline"].update(dominant_from_profile("qr") or {})
                    del st, ov
                    if not os.environ.get("BENCH_KEEP_CACHE"):  # diagnostic
                        torch.cuda.empty_cache()
                except Exception as ex:  # keep the headline line even if an extra fails
                    others[name] = {"error": str(ex)[:200]}
            out["others"] = others
            # which kind of box of the pool produced this line: the latency-bound leaf kernels run 1.25-1.3 x slower
            # on one population at identical product rates; the LU leaf's time per column, measured in this run, tells them apart
            try:
                upc = others["lu_f64_n16384"]["roofline"]["dominant_kernel"]["chain_kernel"]["us_per_column"]
                out["box_kind"] = {"kind": "fast" if upc < 3.4 else "slow", "lu_leaf_us_per_column": upc,
                                   "rule": "LU leaf < 3.4 us per column in situ: the faster population (2.9-3.0), else the slower one (3.8-3.9)"}
            except Exception:
                out["box_kind"] = None

        # ---------------------------------------------------------------- CPU baseline (oracle port, bounded sample)
        if dist is None and not args.no_cpu:
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
            # BASELINE.json configs[0] itself: fp64 matmul + llt on a 1024 x 1024 random SPD Mat (faer/examples/bench.rs:1511-1539) -- the
            # port, the host's OpenBLAS and this GPU on exactly those shapes, best of several repetitions each
            cfg0 = {}
            try:
                n0 = 1024
                h0 = np.asfortranarray(rng.standard_normal((n0, n0)))
                s0 = np.asfortranarray(h0 @ h0.T + n0 * np.eye(n0))  # bench.rs:1511-1513
                c0 = np.zeros((n0, n0), order="F")

                def best_of(fn, reps):
                    tb = 1e30
                    for _ in range(reps):
                        t0_ = time.perf_counter()
                        fn()
                        tb = min(tb, time.perf_counter() - t0_)
                    return tb

                t_pm = best_of(lambda: orc.matmul(c0, h0, s0), 5)
                t_pl = best_of(lambda: orc.llt_in_place(s0.copy(order="F")), 5) - best_of(lambda: s0.copy(order="F"), 5)
                cfg0["port_matmul_GFLOP/s"] = round(2.0 * n0 ** 3 / t_pm / 1e9, 2)
                cfg0["port_llt_GFLOP/s"] = round(n0 ** 3 / 3.0 / max(t_pl, 1e-9) / 1e9, 2)
                try:
                    cfg0["openblas_dgemm_GFLOP/s"] = round(2.0 * n0 ** 3 / best_of(lambda: sblas.dgemm(1.0, h0, s0), 10) / 1e9, 1)
                    cfg0["openblas_dpotrf_GFLOP/s"] = round(n0 ** 3 / 3.0 / best_of(lambda: slap.dpotrf(s0, lower=1), 10) / 1e9, 1)
                except Exception as ex:
                    cfg0["openblas_error"] = str(ex)[:120]
                # the GPU on the same shapes, operands resident in HBM, events on the library's stream, best of 20
                d_a = torch.from_numpy(np.ascontiguousarray(h0.T)).to(dev).t()
                d_s = torch.from_numpy(np.ascontiguousarray(s0.T)).to(dev).t()
                d_c = torch.empty((n0, n0), dtype=torch.float64, device=dev).t()
                d_w = d_s.clone()

                def gpu_best(fn, reps=20):
                    fn()
                    torch.cuda.synchronize()
                    tb = 1e30
                    for _ in range(reps):
                        e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0_.record()
                        fn()
                        e1_.record()
                        torch.cuda.synchronize()
                        tb = min(tb, e0_.elapsed_time(e1_) * 1e-3)
                    return tb

                t_gm = gpu_best(lambda: F.matmul(d_c, F.ACCUM_REPLACE, d_a, d_s, 1.0))

                def gllt():
                    d_w.copy_(d_s)
                    F.llt_factor_in_place(d_w)

                t_gl = gpu_best(gllt) - gpu_best(lambda: d_w.copy_(d_s))
                cfg0["gpu_matmul_GFLOP/s"] = round(2.0 * n0 ** 3 / t_gm / 1e9, 1)
                cfg0["gpu_llt_GFLOP/s"] = round(n0 ** 3 / 3.0 / max(t_gl, 1e-9) / 1e9, 1)
                cfg0["gpu_matmul_us"], cfg0["gpu_llt_us"] = round(t_gm * 1e6, 1), round(t_gl * 1e6, 1)
                cfg0["workload"] = "BASELINE configs[0]: fp64 matmul + llt, 1024 x 1024 random SPD Mat"
                cfg0["note"] = (f"port = oracle/ ({nthr} OpenMP threads); openblas = scipy on the same host; gpu = this library, operands in HBM, "
                                "the llt call includes its one host synchronisation (status word); faer itself cannot be built here")
            except Exception as ex:
                cfg0 = {"error": str(ex)[:200]}
            out["cpu_baseline"] = {"value": round(2.0 * n_mm ** 3 / t_mm / 1e9, 2), "unit": "GFLOP/s", "cores": nthr,
                                   "kind": "port", "config_r_1024": cfg0,
                                   "sample": f"oracle (C restatement of faer's algorithm, OpenMP over columns, {nthr} threads of "
                                             f"{ncpu} host cores) fp64 matmul n={n_mm}, 1 rep ({t_mm:.2f} s); "
                                             f"llt n={n_llt}: {n_llt ** 3 / 3.0 / t_llt / 1e9:.2f} GFLOP/s ({t_llt:.2f} s)",
                                   "note": "faer itself cannot be built here (no Rust toolchain); proxy, see BASELINE.md"}
        try:
            # RCCL prints a version banner through C stdio, which is block buffered on a pipe and would otherwise land BEHIND this
            # line at exit: flush it first, so that the JSON line is the last line of stdout
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
