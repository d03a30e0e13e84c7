"""One-process experiment: XCC placement of the internal streams, LU / LLT timings under the current env switches
("Environment switches").  GPU box only."""
import collections
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
what = sys.argv[1] if len(sys.argv) > 1 else "lu"
if os.environ.get("EXP_LEND") is not None:  # A/B: lending the panel stream's idle CUs to the big products (round 6)
    L.faer_hip_debug_lend_cus(int(os.environ["EXP_LEND"]))
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("FAER_HIP_") or k.startswith("EXP_"))


def timeit(fn, reset, reps=4):
    best = 1e9
    for _ in range(reps):
        reset()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


if what == "xcc":
    for which, name in ((1, "bulk"), (2, "panel")):
        nb = 64
        out = (ctypes.c_uint * (2 * nb))()
        L.faer_hip_debug_stream_xcc(which, nb, out)
        xcc = collections.Counter(out[2 * i] & 0xF for i in range(nb))
        cus = len({(out[2 * i] & 0xF, out[2 * i + 1] & 0xFFFF00) for i in range(nb)})
        print(f"[{tag}] {name}: blocks per XCC {dict(sorted(xcc.items()))}, distinct (xcc, hw_id>>8) = {cus}", flush=True)
elif what == "lu":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    g = torch.Generator(device="cuda").manual_seed(4)
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    work = a.clone()
    res = {}

    def run():
        res["perm"] = F.partial_piv_lu_factor_in_place(work)[0]

    ms = timeit(run, lambda: work.copy_(a))
    p = torch.as_tensor(res["perm"].astype(np.int64), device="cuda")
    Lm = torch.tril(work, -1) + torch.eye(n, dtype=torch.float64, device="cuda")
    x = torch.randn((n, 2), dtype=torch.float64, device="cuda")
    r = (Lm @ (torch.triu(work) @ x) - a[p] @ x).abs().max().item()
    csum = float(work.double().sum().item())
    print(f"[{tag}] lu n={n}: {ms:.2f} ms, residual {r:.2e}, checksum {csum!r}, perm crc {__import__('zlib').crc32(res['perm'].tobytes())}", flush=True)
elif what == "llt":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    g = torch.Generator(device="cuda").manual_seed(5)
    b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    a = (b @ b.t() / n + torch.eye(n, dtype=torch.float64, device="cuda") * 2).t().contiguous().t()
    del b
    work = a.clone()
    ms = timeit(lambda: F.llt_factor_in_place(work), lambda: work.copy_(a))
    Lm = torch.tril(work)
    x = torch.randn((n, 2), dtype=torch.float64, device="cuda")
    r = (Lm @ (Lm.t() @ x) - a @ x).abs().max().item()
    print(f"[{tag}] llt n={n}: {ms:.2f} ms, residual {r:.2e}, checksum {float(Lm.sum().item())!r}", flush=True)
