"""A flat 256-column LU panel on the CU-masked panel stream, alone and beside a product on the bulk stream: what the leaf, the
interchange and the fused node kernels cost per launch (run under rocprofv3 --kernel-trace --stats; PANEL_MODE=alone|busy,
PANEL_ROWS=<m>).  GPU box only."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
mode = os.environ.get("PANEL_MODE", "alone")
m = int(os.environ.get("PANEL_ROWS", "8704"))
w = int(os.environ.get("PANEL_COLS", "256"))
N = 16384
g = torch.Generator(device="cuda").manual_seed(4)
a = torch.randn((512, N), dtype=torch.float64, device="cuda", generator=g).t()  # column major, leading dimension N
work = a.clone()
view = work[N - m:, :w]
L.faer_hip_debug_internal_stream.restype = C.c_void_p
bulk = L.faer_hip_debug_internal_stream(1)
panel = L.faer_hip_debug_internal_stream(2)
n = 8192
x = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
y = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
z = torch.zeros((n, n), dtype=torch.float64, device="cuda").t()
torch.cuda.synchronize()
for rep in range(4):
    work.copy_(a)
    torch.cuda.synchronize()
    if mode == "busy":
        L.faer_hip_set_stream(C.c_void_p(bulk))
        for _ in range(2):
            F.matmul(z, False, x, y)
    L.faer_hip_set_stream(C.c_void_p(panel))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    F.partial_piv_lu_factor_in_place(view)
    torch.cuda.synchronize()
print("done", mode, m, w, flush=True)
if hasattr(L, "faer_hip_debug_dump_timing"):
    L.faer_hip_debug_dump_timing()
