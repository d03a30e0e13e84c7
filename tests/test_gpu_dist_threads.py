"""The stream schedule a rank runs over the BUILT-IN transport (csrc/dist.hip `quiet`: broadcasts start from the stream that packed or
last read the buffer, waits are taken by the internal streams, nothing on the caller's stream between the first step and the end) with
2 and 3 ranks on ONE GPU: RCCL refuses two ranks on a device, so the ranks are threads of one process over the loop-back transport
(csrc/loop_transport.hip -- same contract as the RCCL transport, device-to-device copies).  Results against the CPU oracle: identical
pivots, factors within tolerance; the wire plan (broadcasts per rank) as the host-backend tests expect it."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, threading
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as ge
F = ge.load_package()
what, n, nb, world, out = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
torch.cuda.set_device(0)
F.lib()
g = torch.Generator(device="cuda").manual_seed(4321)
a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
if what == "llt":
    a = (a @ a.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t().contiguous().t()
torch.cuda.synchronize()
grp = F.LoopbackGroup(world)
res, errs = [None] * world, []

def run_rank(rank):
    try:
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()  # (a stream of its own per rank: two ranks behind ONE null stream would wait for each other)
        with torch.cuda.stream(s):
            F.use_torch_stream()
            cols = [c for b in range(rank, (n + nb - 1) // nb, world) for c in range(b * nb, min(n, (b + 1) * nb))]
            loc = a[:, cols].t().contiguous().t()
            loc0 = loc.clone()
            tr = grp.rank(rank)
            fwd = None
            if what == "lu":
                fwd, _, cnt = F.dist_partial_piv_lu(loc, n, nb, rank, world, transport=tr)
            else:
                cnt = F.dist_llt(loc, n, nb, rank, world, transport=tr)
            F.synchronize()
            s.synchronize()
            above = torch.arange(n, device="cuda")[:, None] < torch.tensor(cols, device="cuda")[None, :]
            res[rank] = dict(cols=np.array(cols), loc=loc.cpu().numpy(), fwd=fwd, cnt=cnt, st=tr.stats(),
                             upper_untouched=bool(torch.equal(loc[above], loc0[above])))
            tr.close()
    except BaseException as ex:
        errs.append(f"rank {rank}: {ex!r}")

ths = [threading.Thread(target=run_rank, args=(r,)) for r in range(world)]
for t in ths:
    t.start()
for t in ths:
    t.join()
assert not errs, errs
np.savez(out, a=a.cpu().numpy(), **{f"{k}{r}": (np.array(-1) if v is None else v) for r in range(world) for k, v in res[r].items() if k != "st"},
         **{f"nb{r}": res[r]["st"]["broadcasts"] for r in range(world)})
'''


def run(tmp_path, what, n, nb, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "out.npz"
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    p = subprocess.run([sys.executable, str(script), ROOT, what, str(n), str(nb), str(world), str(out)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    return np.load(out)


@pytest.mark.parametrize("n,nb,world", [(1024, 128, 2), (1500, 256, 2), (1536, 128, 3), (2304, 256, 3), (4096, 512, 2)])
def test_threads_lu_over_the_builtin_schedule(tmp_path, oracle, n, nb, world):
    r = run(tmp_path, "lu", n, nb, world)
    got = np.zeros((n, n))
    for q in range(world):
        got[:, r[f"cols{q}"]] = r[f"loc{q}"]
        assert np.array_equal(r[f"fwd{q}"], r["fwd0"]) and int(r[f"cnt{q}"]) == int(r["cnt0"])
        assert int(r[f"nb{q}"]) == (n + nb - 1) // nb  # every rank takes part in one broadcast per block column
    o = np.asfortranarray(r["a"])
    operm, _, ont = oracle.lu_in_place(o)
    assert np.array_equal(np.asarray(r["fwd0"]).astype(np.int64), operm) and int(r["cnt0"]) == ont
    kappa = np.linalg.cond(r["a"][operm])
    assert np.abs(got - o).max() <= 4 * n * 2.3e-16 * kappa * max(1.0, np.abs(o).max())


@pytest.mark.parametrize("n,nb,world", [(1024, 128, 2), (1000, 192, 2), (1536, 128, 3), (2500, 256, 3), (4096, 512, 2)])
def test_threads_llt_over_the_builtin_schedule(tmp_path, oracle, n, nb, world):
    r = run(tmp_path, "llt", n, nb, world)
    got = np.zeros((n, n))
    for q in range(world):
        got[:, r[f"cols{q}"]] = r[f"loc{q}"]
        assert int(r[f"cnt{q}"]) == 0 and bool(r[f"upper_untouched{q}"])
    o = np.asfortranarray(r["a"])
    assert oracle.llt_in_place(o) == ("ok", 0)
    il = np.tril_indices(n)
    kappa = np.linalg.cond(r["a"])
    assert np.abs(got[il] - o[il]).max() <= 8 * n * 2.3e-16 * kappa * np.abs(o[il]).max()
