"""The in-situ tax of the trailing products, itemised (VERDICT r05 item 1a -> profiles/r06_update_in_situ.txt).

One LU / LLT at N = 16384 is run with the library's per-launch profile (faer_hip_prof_end_spans): every big-tile product comes back
with its shape (m, n, k, Lower?), its stream, its start and its duration INSIDE the factorization.  Every distinct shape is then
replayed ALONE (a) on the bulk stream (the 224 CUs it had in the factorization) and (b) on the whole chip, and the table puts the
three next to each other together with what a launch loses to tile quantisation on its own (tiles / (2 workgroups x CUs) rounded up).
GPU box only."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

import ctypes as C  # noqa: E402

F = ge.load_package()
L = F.lib()
torch.cuda.set_device(0)
F.use_torch_stream()
what = sys.argv[1] if len(sys.argv) > 1 else "lu"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
PEAK = 78.6


def cm(m, k):
    return torch.randn((k, m), dtype=torch.float64, device="cuda").t()


g = torch.Generator(device="cuda").manual_seed(4)
if what == "lu":
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    run = lambda w: F.partial_piv_lu_factor_in_place(w)  # noqa: E731
else:
    b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    a = (b @ b.t() / n + torch.eye(n, dtype=torch.float64, device="cuda") * 2).t().contiguous().t()
    del b
    run = lambda w: F.llt_factor_in_place(w)  # noqa: E731
work = a.clone()
for _ in range(2):
    work.copy_(a)
    run(work)
torch.cuda.synchronize()
# wall time of the unprofiled call
best = 1e9
for _ in range(3):
    work.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(work)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
work.copy_(a)
torch.cuda.synchronize()
F.prof_begin()
run(work)
tot, spans = F.prof_end_spans()
prods = [s for s in spans if s["cls"] == "mfma_products"]
del work, a
torch.cuda.empty_cache()

L.faer_hip_debug_internal_stream.restype = C.c_void_p
bulk_ptr = L.faer_hip_debug_internal_stream(1)
bulk = torch.cuda.ExternalStream(bulk_ptr)
cur = torch.cuda.current_stream()


def replay(m_, n_, k_, tri, stream_ptr, stream):
    """the same product alone on `stream`: milliseconds (best of 3 pairs)"""
    lower = tri > 0
    x = cm(m_, k_)
    if lower:
        c = cm(m_, m_)
        y = x
    else:
        c = cm(m_, n_)
        y = cm(k_, n_) if what == "lu" else cm(n_, k_).t()
    torch.cuda.synchronize()
    L.faer_hip_set_stream(C.c_void_p(stream_ptr))
    best_ = 1e9
    try:
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if lower:
                F.gemm(c, F.DST_LOWER, F.ACCUM_ADD, x, y.t(), -1.0)
            else:
                F.gemm(c, F.DST_FULL, F.ACCUM_ADD, x, y, -1.0)
            e1.record(stream)
            torch.cuda.synchronize()
            if rep:
                best_ = min(best_, e0.elapsed_time(e1))
    finally:
        F.use_torch_stream()
    if lower and tri > 1:  # the in-situ launch skipped the first tri - 1 rows of the triangle: same rate, its own flops
        full = 2.0 * k_ * 0.5 * m_ * (m_ + 1)
        skip = 2.0 * k_ * 0.5 * (tri - 1) * tri
        best_ *= (full - skip) / full
    return best_


def tiles(m_, n_, tri):
    tm, tn = math.ceil(m_ / 128), math.ceil(n_ / 128)
    if tri > 0:
        sk = (tri - 1) // 128
        return tm * (tm + 1) // 2 - sk * (sk + 1) // 2
    return tm * tn


cache = {}
print(f"# {what} n={n}: wall {best:.2f} ms unprofiled; {len(prods)} big-tile products, {tot['mfma_products']['ms']:.2f} ms inside them "
      f"({tot['mfma_products']['units'] / tot['mfma_products']['ms'] / 1e9:.1f} TFLOP/s in situ = {tot['mfma_products']['units'] / tot['mfma_products']['ms'] / 1e9 / PEAK:.3f} of peak)")
print("# t0_ms  stream      m      n     k  lower  tiles  waves@448 quant_eff | in_situ_ms  TF/s | alone_224cu_ms  TF/s | alone_256cu_ms  TF/s | "
      "in_situ/alone224  alone224/ideal224")
sum_in = sum_224 = sum_256 = sum_quant = 0.0
ideal224 = 0.0
for s in prods:
    m_, n_, k_, tri = s["d"]
    key = (m_, n_, k_, tri)
    if key not in cache:
        t224 = replay(m_, n_, k_, tri, bulk_ptr, bulk)
        t256 = replay(m_, n_, k_, tri, cur.cuda_stream, cur)
        cache[key] = (t224, t256)
    t224, t256 = cache[key]
    nt = tiles(m_, n_, tri)
    waves = nt / 448.0
    qe = waves / math.ceil(waves)
    fl = s["units"]
    sum_in += s["ms"]
    sum_224 += t224
    sum_256 += t256
    sum_quant += t224 * qe
    # the standalone whole-chip rate of a large K = k product scaled to 224 CUs: what the launch would take without quantisation or neighbours
    print(f"{s['t0_ms']:7.2f}  {s['stream']:6s} {m_:6d} {n_:6d} {k_:5d}  {int(tri > 0):5d} {nt:6d}  {waves:8.2f}  {qe:8.3f} | {s['ms']:9.3f} {fl / s['ms'] / 1e9:6.1f} | "
          f"{t224:9.3f} {fl / t224 / 1e9:6.1f} | {t256:9.3f} {fl / t256 / 1e9:6.1f} | {s['ms'] / t224:8.3f}")
fl_all = tot["mfma_products"]["units"]
print(f"# sums: in situ {sum_in:.2f} ms ({fl_all / sum_in / 1e9:.1f} TF/s) | alone on the bulk stream's 224 CUs {sum_224:.2f} ms ({fl_all / sum_224 / 1e9:.1f} TF/s) | "
      f"alone on 256 CUs {sum_256:.2f} ms ({fl_all / sum_256 / 1e9:.1f} TF/s)")
print(f"# tax of the neighbours (in situ - alone on 224 CUs): {sum_in - sum_224:.2f} ms = {(sum_in / sum_224 - 1) * 100:.1f} %")
print(f"# tile quantisation inside the alone-224 time (partial last wave of 448 slots): {sum_224 - sum_quant:.2f} ms = {(1 - sum_quant / sum_224) * 100:.1f} %")
print(f"# the reserved CUs (224 instead of 256): {sum_224 - sum_256:.2f} ms")
