"""Per-tile phase accounting of the pipelined GEMM kernel (timing build: make -C faer-rs_amd/csrc timing;
FAER_HIP_LIB=faer-rs_amd/libfaer_hip_timing.so python tools/gpu_gemm_phases.py).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()


def cm(m, n):
    return torch.randn((n, m), dtype=torch.float64, device="cuda").t()


def run(label, fn, reps=4):
    fn()
    torch.cuda.synchronize()
    L.faer_hip_debug_dump_timing()  # reset
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) / reps:.3f} ms", file=sys.stderr, flush=True)
    L.faer_hip_debug_dump_timing()


r = 15360
for k in (512, 1024, 4096):
    x, y, c = cm(r, k), cm(r, k), cm(r, r)
    run(f"full  r={r} k={k} accumulate", lambda: F.gemm(c, F.DST_FULL, F.ACCUM_ADD, x, y.t(), -1.0))
    run(f"lower r={r} k={k} accumulate", lambda: F.gemm(c, F.DST_LOWER, F.ACCUM_ADD, x, x.t(), -1.0))
    run(f"full  r={r} k={k} replace", lambda: F.gemm(c, F.DST_FULL, F.ACCUM_REPLACE, x, y.t(), 1.0))
    del x, y, c
