"""Two ranks of the distributed drivers with the DEVICE backend (csrc/dist.hip) sharing one GPU: the multi-rank
control flow (ownership, look-ahead order, two alternating panel buffers, status exchange) on real device kernels.
The transport is gloo on host copies of the device buffer (one GPU box, no RCCL ring to form); the results must
equal the single-GPU factorization of the same matrix by the same library AND the CPU oracle's (VERDICT r04 weak 11:
a device-backend result compared with the oracle directly, not only transitively): identical pivots, factors within
tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as ge
F = ge.load_package()
what, n, nb, out_dir, use_async = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6] == "1"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
F.lib()
F.use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(1234)
a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
if what == "llt":
    a = (a @ a.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t().contiguous().t()
cols = [c for b in range(rank, (n + nb - 1) // nb, world) for c in range(b * nb, min(n, (b + 1) * nb))]
loc = a[:, cols].t().contiguous().t()

def bcast(t, root):
    torch.cuda.current_stream().synchronize()
    h = t.cpu()
    dist.broadcast(h, src=root)
    t.copy_(h)

class Done:
    def wait(self):
        pass

def ibcast(t, root):
    bcast(t, root)
    return Done()

if what == "lu":
    fwd, bwd, cnt = F.dist_partial_piv_lu(loc, n, nb, rank, world, bcast, ibcast=ibcast if use_async else None)
    F.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cols=np.array(cols), loc=loc.cpu().numpy(), fwd=fwd, cnt=cnt)
else:
    loc0 = loc.clone()
    cnt = F.dist_llt(loc, n, nb, rank, world, bcast, ibcast=ibcast if use_async else None)
    F.synchronize()
    # only the lower triangle of the global matrix may be written (the grouped update masks a staircase)
    above = torch.arange(n, device="cuda")[:, None] < torch.tensor(cols, device="cuda")[None, :]
    upper_untouched = bool(torch.equal(loc[above], loc0[above]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cols=np.array(cols), loc=loc.cpu().numpy(), cnt=cnt, upper_untouched=upper_untouched)
if rank == 0:  # single-GPU reference by the same library
    ref = a.clone()
    if what == "lu":
        p, _, c = F.partial_piv_lu_factor_in_place(ref)
        np.savez(os.path.join(out_dir, "ref.npz"), ref=ref.cpu().numpy(), fwd=p, cnt=c, a=a.cpu().numpy())
    else:
        c = F.llt_factor_in_place(ref)
        np.savez(os.path.join(out_dir, "ref.npz"), ref=ref.cpu().numpy(), cnt=c, a=a.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
'''


def run(tmp_path, what, n, nb, use_async):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 33000 + (os.getpid() + n + nb) % 2000
    procs = []
    for r in range(2):
        # FAER_HIP_DIST_TWO_MIN=0: every step of the LU on the bulk + panel streams (by default steps with little trailing
        # work run on the caller's stream alone), so the two-stream schedule is what the two ranks exercise
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FAER_HIP_DIST_TWO_MIN="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, what, str(n), str(nb), str(tmp_path), "1" if use_async else "0"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(2)], np.load(tmp_path / "ref.npz")


@pytest.mark.parametrize("n,nb,use_async", [(1024, 128, True), (900, 256, False)])
def test_two_rank_lu_equals_single_gpu(tmp_path, oracle, n, nb, use_async):
    res, ref = run(tmp_path, "lu", n, nb, use_async)
    got = np.zeros((n, n))
    for r in res:
        assert np.array_equal(r["fwd"], ref["fwd"]) and int(r["cnt"]) == int(ref["cnt"])
        got[:, r["cols"]] = r["loc"]
    assert np.abs(got - ref["ref"]).max() <= 256 * n * 2.3e-16 * max(1.0, np.abs(ref["ref"]).max())
    # ... and the oracle's: same permutation, factors within the forward error of the factorization
    o = np.asfortranarray(ref["a"])
    operm, _, ont = oracle.lu_in_place(o)
    assert np.array_equal(np.asarray(res[0]["fwd"]).astype(np.int64), operm) and int(res[0]["cnt"]) == ont
    kappa = np.linalg.cond(ref["a"][operm])
    assert np.abs(got - o).max() <= 4 * n * 2.3e-16 * kappa * max(1.0, np.abs(o).max())


@pytest.mark.parametrize("n,nb,use_async", [(1024, 128, True), (1000, 192, False), (2500, 128, True)])
def test_two_rank_llt_equals_single_gpu(tmp_path, oracle, n, nb, use_async):
    res, ref = run(tmp_path, "llt", n, nb, use_async)
    got = np.zeros((n, n))
    for r in res:
        assert int(r["cnt"]) == 0
        assert bool(r["upper_untouched"])
        got[:, r["cols"]] = r["loc"]
    il = np.tril_indices(n)
    assert np.abs(got[il] - ref["ref"][il]).max() <= 256 * n * 2.3e-16 * np.abs(ref["ref"][il]).max()
    # ... and the oracle's Cholesky factor of the same matrix
    o = np.asfortranarray(ref["a"])
    assert oracle.llt_in_place(o) == ("ok", 0)
    kappa = np.linalg.cond(ref["a"])
    assert np.abs(got[il] - o[il]).max() <= 8 * n * 2.3e-16 * kappa * np.abs(o[il]).max()
