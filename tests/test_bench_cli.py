"""bench.py's command-line contract (no GPU): what each invocation runs and what a distributed line carries.

VERDICT r04 item 4: `bench.py --gpus N` (N > 1) without --workload must run BASELINE.json configs[3] -- the 1-D block-cyclic
partial-pivot LU over the library's RCCL transport, strong scaling, with `per_rank.ncclCommCount` on the line -- and not the
collective-free GEMM; `--gpus 1` stays the DGEMM of configs[1]."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_default_workload_one_gpu_is_the_dgemm():
    assert bench.resolve_run(1, None, None) == ("gemm", "torch", True)


def test_default_workload_several_gpus_is_the_block_cyclic_lu_over_rccl():
    for n in (2, 4, 8):
        wl, tr, defaulted = bench.resolve_run(n, None, None)
        assert (wl, tr, defaulted) == ("lu", "rccl", True)


def test_explicit_choices_are_kept():
    assert bench.resolve_run(8, "gemm", None) == ("gemm", "torch", False)
    assert bench.resolve_run(8, "llt", "rccl") == ("llt", "rccl", False)
    assert bench.resolve_run(2, "lu", None) == ("lu", "rccl", False)  # (round 6: the library's own transport for every distributed factorization)
    assert bench.resolve_run(2, "llt", None) == ("llt", "rccl", False) and bench.resolve_run(2, "lu", "torch") == ("lu", "torch", False)
    assert bench.resolve_run(1, "lu", "rccl") == ("lu", "rccl", False)


def test_per_rank_carries_ncclcommcount_of_every_rank():
    world = 4
    allr = np.array([[101.0, 40.0, 8, world, 32, 2.0e9, 3.5]] * world)
    pr = bench.per_rank_dict(allr, "rccl")
    assert pr["ncclCommCount"] == [world] * world and pr["transport"] == "rccl"
    assert pr["update_and_wait_device_ms"] == [61.0] * world and pr["panels_owned"] == [8] * world


def test_source_wires_the_defaults_into_the_run():
    """the distributed branch is what the resolved names select: lu x world > 1 -> dist_partial_piv_lu with the RCCL transport,
    scaling strong, per_rank on the line"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "args.workload, args.transport, defaulted = resolve_run(args.gpus, args.workload, args.transport)" in src
    assert re.search(r'rccl = F\.RcclTransport\(', src) and "transport=rccl" in src
    assert '"scaling": "strong" if (args.workload in ("lu", "llt") and dist is not None) else "weak"' in src
    assert 'out["per_rank"] = per_rank' in src and "lu_1gpu_same_run" in src
