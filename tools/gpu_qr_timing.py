"""one QR of 5e5 x 256 fp32 with the instrumented library (make -C faer-rs_amd/csrc timing; FAER_HIP_LIB=faer-rs_amd/libfaer_hip_timing.so):
prints the phase accounting of the one-pass path's panel kernel (shader cycles, tsqr.hip TQ_STAMP)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as ge

F = ge.load_package()
F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
m, n = 500000, 256
g = torch.Generator(device="cuda").manual_seed(5)
a = torch.randn((n, m), dtype=torch.float32, device="cuda", generator=g).t()
h = torch.zeros((n, n), dtype=torch.float32, device="cuda").t()
for _ in range(2):
    w = a.clone()
    assert F.qr_factor_in_place(w, h) == n
F.synchronize()
