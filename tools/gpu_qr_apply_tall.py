"""tall-skinny least squares pieces: qr_factor_in_place against applying Q^T / Q to right-hand sides (householder.rs:724-808)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
F = ge.load_package(); torch.cuda.set_device(0); F.lib(); F.use_torch_stream()
def best_of(fn, reps=4):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
for dt in (torch.float64, torch.float32):
    for m, n in [(500000, 256), (100000, 512), (20000, 128)]:
        g = torch.Generator(device="cuda").manual_seed(1)
        a = torch.randn((n, m), dtype=dt, device="cuda", generator=g).t()
        bs = int(F.qr_recommended_block_size(m, n, "float64" if dt == torch.float64 else "float32"))
        w = a.clone(); h = torch.zeros((n, bs), dtype=dt, device="cuda").t()
        tq = best_of(lambda: (w.copy_(a), F.qr_factor_in_place(w, h)))
        tc = best_of(lambda: w.copy_(a))
        F.qr_factor_in_place(w, h)
        out = [f"{str(dt)[6:]} {m} x {n} bs {bs}: factor {tq - tc:.2f} ms;"]
        for k in (1, 16, 256):
            b = torch.randn((k, m), dtype=dt, device="cuda", generator=g).t()
            t1 = best_of(lambda: F.apply_block_householder_sequence_on_the_left_in_place(w, h, b, transpose=True))
            t2 = best_of(lambda: F.apply_block_householder_sequence_on_the_left_in_place(w, h, b, transpose=False))
            out.append(f"Q^T B / Q B with {k} rhs {t1:.2f} / {t2:.2f} ms;")
        print(" ".join(out), flush=True)
