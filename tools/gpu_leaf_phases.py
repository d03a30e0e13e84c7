"""Phase accounting of the TRSM substitution leaf (timing build: make -C faer-rs_amd/csrc timing).
usage: FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_leaf_phases.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
for dt in (torch.float64, torch.float32):
    for n, k in [(128, 64), (128, 8192), (128, 16384)]:
        t = (torch.tril(torch.randn((n, n), dtype=dt, device="cuda")) / n + torch.eye(n, dtype=dt, device="cuda")).t().contiguous().t()
        for side, x in (("left (through the tile)", torch.randn((k, n), dtype=dt, device="cuda").t()),
                        ("right (lanes along rhs)", torch.randn((n, k), dtype=dt, device="cuda"))):
            for _ in range(10):
                F.solve_lower_triangular_in_place(t, x)
            torch.cuda.synchronize()
            print(f"{str(dt)[6:]} n={n} k={k} {side}:", file=sys.stderr, flush=True)
            L.faer_hip_debug_dump_timing()
