"""A/B of the dense kernel's tile shapes on one visit: DGEMM N = 8192 (bench config G) per ctx().gemm_variant, with a
checksum against torch (independent of the variant)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as ge

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
c = torch.empty((n, n), dtype=torch.float64, device="cuda").t()
x = torch.randn((n,), dtype=torch.float64, device="cuda", generator=g)
ref = a @ (b @ x)
for v in [int(t) for t in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,3,4,0".split(","))]:
    L.faer_hip_set_gemm_variant(v)
    for _ in range(2):
        F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 8
    for _ in range(reps):
        F.matmul(c, F.ACCUM_REPLACE, a, b, 1.0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    err = ((c @ x) - ref).abs().max().item() / ref.abs().max().item()
    print(f"variant {v}: {ms:.3f} ms = {2.0 * n ** 3 / ms / 1e9:.1f} TFLOP/s, checksum rel err {err:.2e}", flush=True)
L.faer_hip_set_gemm_variant(0)
