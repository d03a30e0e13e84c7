"""Host time of one broadcast of the library's RCCL transport with a ONE-rank communicator (what the one-rank dry runs of the
distributed drivers pay per panel / chunk): ibcast + wait through the FaerHipComm function pointers, by message size."""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

F = ge.load_package()
torch.cuda.set_device(0)
F.lib()
F.use_torch_stream()
tr = F.RcclTransport(F.RcclTransport.unique_id(), 0, 1)
comm = tr.comm
IB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int)
WT = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
ib = comm.ibcast
wt = comm.wait
for mb in (0.001, 1, 16, 64):
    buf = torch.zeros(int(mb * 1e6), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(100):
            ib(comm.user, buf.data_ptr(), buf.numel(), 0, i % 8)
        t1 = time.perf_counter()
        for i in range(8):
            wt(comm.user, i)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
    print(f"{mb:8.3f} MB: ibcast host {1e4 * (t1 - t0):7.1f} us per call, 8 waits {1e6 * (t2 - t1):7.1f} us, drain {1e3 * (t3 - t2):.2f} ms")
# does the call block the host while the stream it is ordered behind is busy?  (a ~20 ms product in front)
a = torch.randn(8192, 8192, dtype=torch.float64, device="cuda")
buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    c = a @ a
    t1 = time.perf_counter()
    ib(comm.user, buf.data_ptr(), buf.numel(), 0, 0)
    t2 = time.perf_counter()
    wt(comm.user, 0)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"behind a busy stream: launch {1e6 * (t1 - t0):.0f} us, ibcast host {1e6 * (t2 - t1):.0f} us, wait host {1e6 * (t3 - t2):.0f} us, drain {1e3 * (t4 - t3):.2f} ms")
