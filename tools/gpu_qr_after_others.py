"""QR 5e5 x 256 timed alone, then again after LLT runs of N = 16384 in the same process -- with the same tensors, with tensors
allocated afterwards, with the block size bench.py uses (what state makes `others.qr_f32_500000x256` of bench.py slower
than `bench.py --workload qr`?)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as ge

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
m, n = 500000, 256
g = torch.Generator(device="cuda").manual_seed(5)


def make(bs):
    a = torch.randn((n, m), dtype=torch.float32, device="cuda", generator=g).t()
    work = a.clone()
    h = torch.zeros((n, bs), dtype=torch.float32, device="cuda").t()
    return a, work, h


def qr_ms(t, reps=5):
    a, work, h = t
    for _ in range(2):
        work.copy_(a)
        F.qr_factor_in_place(work, h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        work.copy_(a)
        F.qr_factor_in_place(work, h)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        work.copy_(a)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return round(((t1 - t0) - (t2 - t1)) / reps * 1e3, 3)


bs = F.qr_recommended_block_size(m, n, np.float32)
print("recommended block size", bs)
t256, tbs = make(256), make(bs)
print("fresh: bs=256", qr_ms(t256), qr_ms(t256), " bs=rec", qr_ms(tbs), qr_ms(tbs))
N = 16384
b = torch.randn((N, N), dtype=torch.float64, device="cuda", generator=g).t()
spd = (b @ b.t() + N * torch.eye(N, dtype=torch.float64, device="cuda")).t().contiguous().t()
w = spd.clone()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 7):
    w.copy_(spd)
    F.llt_factor_in_place(w)
torch.cuda.synchronize()
print("after llt x7, old tensors: bs=256", qr_ms(t256), qr_ms(t256), " bs=rec", qr_ms(tbs), qr_ms(tbs))
del b, spd, w
torch.cuda.empty_cache()
t2 = make(bs)
print("new tensors after llt: bs=rec", qr_ms(t2), qr_ms(t2), " old again:", qr_ms(tbs))
