"""world_size-2 (and 3) CPU tests of the distributed LU orchestration (csrc/dist_lu.h) over gloo.

The GPU product path instantiates the same template with the device backend (csrc/dist.hip); here it runs with
the test-only host backend of tests/dist_host_backend.cpp, one process per rank, torch.distributed broadcasts.
Checked against the single-process oracle: identical pivots, factors within tolerance, ONE broadcast per block
column with exactly {pivots, panel} bytes."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
LIB = os.path.join(BUILD, "libdist_host_test.so")


def build_lib():
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "tests", "dist_host_backend.cpp")
    hdrs = [os.path.join(ROOT, "faer-rs_amd", "csrc", h) for h in ("dist_lu.h", "dist_llt.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", LIB, src])
    return LIB


WORKER = r'''
import ctypes as C, os, sys
import numpy as np
import torch
import torch.distributed as dist

root, lib_path, m, n, nb, seed, out_dir = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = C.CDLL(lib_path)
lib.test_dist_lu_f64.restype = C.c_long
lib.test_dist_local_ncols.restype = C.c_long
rng = np.random.default_rng(seed)
a = np.asfortranarray(rng.standard_normal((m, n)))
# this rank's block columns, in global order
cols = [c for b in range(rank, (n + nb - 1) // nb, world) for c in range(b * nb, min(n, (b + 1) * nb))]
assert len(cols) == lib.test_dist_local_ncols(C.c_long(n), C.c_long(nb), rank, world)
a_loc = np.asfortranarray(a[:, cols]) if cols else np.zeros((m, 0), order="F")

BCAST = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
def bcast(user, buf, nbytes, root_rank):
    arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(nbytes,))
    t = torch.from_numpy(arr)
    dist.broadcast(t, src=root_rank)
cb = BCAST(bcast)
piv = np.zeros(min(m, n), dtype=np.int32)
stats = (C.c_ulonglong * 4)()
nbc = lib.test_dist_lu_f64(a_loc.ctypes.data_as(C.c_void_p), C.c_long(m), C.c_long(a_loc.shape[1]), C.c_long(max(m, 1)), C.c_long(n), C.c_long(nb),
                           rank, world, cb, None, piv.ctypes.data_as(C.c_void_p), stats)
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cols=np.array(cols, dtype=np.int64), a_loc=a_loc, piv=piv, nbc=nbc, bytes=stats[0],
         begun=stats[1], waited=stats[2], slot_skew=stats[3])
dist.barrier()
dist.destroy_process_group()
'''


def run_world(tmp_path, world, m, n, nb, seed):
    lib = build_lib()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() + seed) % 2000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, lib, str(m), str(n), str(nb), str(seed), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("world,m,n,nb", [(2, 96, 96, 16), (2, 100, 100, 16), (3, 90, 70, 8), (2, 64, 64, 64), (2, 50, 80, 16)])
def test_dist_lu_matches_single_process_oracle(tmp_path, oracle, world, m, n, nb):
    seed = 7
    res = run_world(tmp_path, world, m, n, nb, seed)
    rng = np.random.default_rng(seed)
    a = np.asfortranarray(rng.standard_normal((m, n)))
    ref = a.copy(order="F")
    perm, perm_inv, _ = oracle.lu_in_place(ref)
    size = min(m, n)
    # every rank holds the same absolute pivots; they reproduce the oracle's permutation
    piv = res[0]["piv"]
    for r in res[1:]:
        assert np.array_equal(r["piv"], piv)
    p = np.arange(m)
    for j in range(size):
        p[[j, piv[j]]] = p[[piv[j], j]]
    assert np.array_equal(p, perm)
    # gather the block columns back and compare the factors
    got = np.zeros((m, n), order="F")
    for r in res:
        if len(r["cols"]):
            got[:, r["cols"]] = r["a_loc"]
    assert np.abs(got - ref).max() <= 200 * max(m, n) * np.finfo(np.float64).eps * max(1.0, np.abs(ref).max())
    # exactly ONE broadcast per factored block column, carrying {pivots, panel}
    nblk = (size + nb - 1) // nb
    hdr = ((nb * 4 + 7) // 8) * 8
    expect_bytes = sum(hdr + (m - k * nb) * min(nb, size - k * nb) * 8 for k in range(nblk))
    for r in res:
        assert int(r["nbc"]) == nblk and int(r["bytes"]) == expect_bytes
        # look-ahead protocol: every broadcast begun once and awaited once, the two buffer slots used alternately
        assert int(r["begun"]) == nblk and int(r["waited"]) == nblk and int(r["slot_skew"]) <= 1
