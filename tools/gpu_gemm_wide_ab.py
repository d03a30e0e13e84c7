"""A/B of the 128 x 256 eight-wavefront tile against the 128 x 128 tile on accumulate-mode updates (the trailing updates
of the factorizations): r x r x K, full and lower, per K"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as ge

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()


def cm(m, n):
    return torch.randn((n, m), dtype=torch.float64, device="cuda").t()


def bench(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for r in (15360, 8192):
    c = cm(r, r)
    for k in (512, 1024, 2048, 4096, 8192):
        x, y = cm(r, k), cm(k, r)
        row = [f"r={r} k={k}:"]
        for kind, name in ((F.DST_FULL, "full"), (F.DST_LOWER, "lower")):
            for v in (5, 3):
                L.faer_hip_set_gemm_variant(v)
                bb = y if kind == F.DST_FULL else x.t()
                ms = bench(lambda: F.gemm(c, kind, F.ACCUM_ADD, x, bb, -1.0))
                fl = 2.0 * r * r * k * (1.0 if kind == F.DST_FULL else 0.5 * (1 + 128.0 / r))
                row.append(f"{name} {'wide' if v == 3 else '128 '} {fl / ms / 1e9:5.1f} TF")
        print("  ".join(row), flush=True)
        del x, y
L.faer_hip_set_gemm_variant(0)
