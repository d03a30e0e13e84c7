"""world_size-2 (and 3) CPU tests of the distributed Cholesky orchestration (csrc/dist_llt.h) over gloo.

The GPU product path instantiates the same template with the device backend (csrc/dist.hip); here it runs with the
test-only host backend of tests/dist_host_backend.cpp, one process per rank, torch.distributed broadcasts.  Checked
against the single-process oracle: the gathered lower triangle within tolerance, an untouched strict upper
triangle, the chunked transfers of every block column -- up to four row chunks of the rows below the diagonal block --
with exactly the planned message count and bytes (+ the closing status exchange), the same failure index on every rank
for a matrix that is not positive definite."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_dist_lu import ROOT, build_lib

WORKER = r'''
import ctypes as C, os, sys
import numpy as np
import torch
import torch.distributed as dist

lib_path, n, nb, seed, bad, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = C.CDLL(lib_path)
lib.test_dist_llt_f64.restype = C.c_long
rng = np.random.default_rng(seed)
b = rng.standard_normal((n, n))
a = np.asfortranarray(b @ b.T + n * np.eye(n))
if bad >= 0:
    a[bad, bad] = -1.0
a[np.triu_indices(n, 1)] = -7.5  # the strict upper triangle must never be read or written
cols = [c for blk in range(rank, (n + nb - 1) // nb, world) for c in range(blk * nb, min(n, (blk + 1) * nb))]
a_loc = np.asfortranarray(a[:, cols]) if cols else np.zeros((n, 0), order="F")

BCAST = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
def bcast(user, buf, nbytes, root_rank):
    arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(nbytes,))
    dist.broadcast(torch.from_numpy(arr), src=root_rank)
cb = BCAST(bcast)
stats = (C.c_ulonglong * 8)()
r = lib.test_dist_llt_f64(a_loc.ctypes.data_as(C.c_void_p), C.c_long(n), C.c_long(a_loc.shape[1]), C.c_long(max(n, 1)), C.c_long(nb),
                          rank, world, cb, None, stats)
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cols=np.array(cols, dtype=np.int64), a_loc=a_loc, ret=r, bytes=stats[0], nbc=stats[1],
         begun=stats[2], waited=stats[3], stray=stats[4], wire_messages=stats[5], wire_bytes=stats[6])
dist.barrier()
dist.destroy_process_group()
'''


def run_world(tmp_path, world, n, nb, seed, bad=-1):
    lib = build_lib()
    script = tmp_path / "worker_llt.py"
    script.write_text(WORKER)
    port = 31500 + (os.getpid() + seed + n) % 2000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), lib, str(n), str(nb), str(seed), str(bad), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("world,n,nb", [(2, 96, 16), (2, 100, 16), (3, 90, 8), (2, 64, 64), (3, 50, 16), (2, 7, 16), (4, 130, 16), (3, 200, 32)])
def test_dist_llt_matches_single_process_oracle(tmp_path, oracle, world, n, nb):
    seed = 11
    res = run_world(tmp_path, world, n, nb, seed)
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((n, n))
    a = np.asfortranarray(b @ b.T + n * np.eye(n))
    ref = a.copy(order="F")
    assert oracle.llt_in_place(ref) == ("ok", 0)
    got = np.zeros((n, n), order="F")
    for r in res:
        assert int(r["ret"]) == 0
        if len(r["cols"]):
            got[:, r["cols"]] = r["a_loc"]
    il = np.tril_indices(n)
    iu = np.triu_indices(n, 1)
    assert (got[iu] == -7.5).all()
    assert np.abs(got[il] - ref[il]).max() <= 200 * n * np.finfo(np.float64).eps * np.abs(ref[il]).max()
    # The rows below every diagonal block travel in up to four block-aligned row chunks (dist_llt.h, LLT_NCH), the diagonal
    # block stays with its owner: the chunk count and the bytes on the wire are those of the plan -- recomputed here from the
    # sizes alone -- plus the closing status exchange (world x 16 bytes); every chunk begun exactly once on every rank,
    # awaited at least once (the look-ahead part and the rest of the update each wait for the chunks they read), none left
    # unawaited, no wait for a transfer that was never started
    nblk = (n + nb - 1) // nb
    messages, payload = 0, 0
    for k in range(nblk):
        tb = nblk - k - 1
        nch = min(4, tb)
        messages += nch
        if nch:
            payload += (n - (k + 1) * nb) * min(nb, n - k * nb) * 8
    for r in res:
        assert int(r["wire_messages"]) == messages and int(r["wire_bytes"]) == payload
        assert int(r["nbc"]) == messages + world and int(r["bytes"]) == payload + world * 16
        assert int(r["begun"]) == messages and int(r["waited"]) >= messages and int(r["stray"]) == 0
    if nblk >= 6:
        assert messages >= 4 * (nblk - 4)  # four chunks per panel while four block rows remain below it


@pytest.mark.parametrize("world,n,nb,bad", [(2, 80, 16, 37), (3, 70, 8, 0), (2, 64, 16, 63)])
def test_dist_llt_failure_index_is_global_and_identical(tmp_path, world, n, nb, bad):
    res = run_world(tmp_path, world, n, nb, 3, bad=bad)
    for r in res:
        assert int(r["ret"]) == -(bad + 1)
