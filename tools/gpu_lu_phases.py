"""Phase accounting of the LU panel kernel (timing build: make -C faer-rs_amd/csrc timing).
usage: FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_lu_phases.py [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cuda").manual_seed(4)
a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
work = a.clone()
for rep in range(3):
    work.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    F.partial_piv_lu_factor_in_place(work)
    e1.record()
    torch.cuda.synchronize()
    print(f"lu n={n} (timing build): {e0.elapsed_time(e1):.2f} ms", file=sys.stderr, flush=True)
    L.faer_hip_debug_dump_timing()
