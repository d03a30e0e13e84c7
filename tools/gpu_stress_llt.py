"""Determinism / race stress of the look-ahead LLT: the same 16384^2 factorization many times, interleaved with
unrelated GPU work, compared BITWISE with the first result; reports the first differing 128-block."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
use_side_stream = len(sys.argv) > 1 and sys.argv[1] == "side"
side = torch.cuda.Stream() if use_side_stream else None
n = 16384
g = torch.Generator(device="cuda").manual_seed(3)
a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
spd = (a @ a.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t()
del a
big = torch.randn((8192, 8192), dtype=torch.float64, device="cuda")
ref = None
bad = 0
ctx = torch.cuda.stream(side) if side is not None else torch.cuda.stream(torch.cuda.current_stream())
torch.cuda.synchronize()  # the inputs were produced on the default stream: a side stream must not start before they exist
with ctx:
    F.use_torch_stream()
    for it in range(24):
        work = spd.clone()
        if it % 3 == 1:
            _ = big @ big  # unrelated work queued right before
        if it % 4 == 2:
            torch.cuda.empty_cache()
        try:
            F.llt_factor_in_place(work)
        except Exception as ex:  # noqa: BLE001
            print(f"iter {it}: EXCEPTION {ex}")
            bad += 1
            continue
        F.synchronize()
        if ref is None:
            ref = work.clone()
            L = torch.tril(ref)
            x = torch.randn((n, 2), dtype=torch.float64, device="cuda")
            r = (L @ (L.t() @ x) - spd @ x).abs().max().item() / (spd.abs() @ x.abs()).max().item()
            print(f"iter 0: relative residual {r:.2e}")
        else:
            d = (torch.tril(work) != torch.tril(ref))
            if d.any().item():
                idx = d.nonzero()[0]
                print(f"iter {it}: MISMATCH, {int(d.sum().item())} entries, first at row {int(idx[0])} col {int(idx[1])} "
                      f"(block {int(idx[0]) // 128}, {int(idx[1]) // 128})")
                bad += 1
print("stress done, bad iterations:", bad, "(side stream)" if use_side_stream else "(default stream)")
