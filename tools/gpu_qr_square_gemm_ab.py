"""square QR: the library's automatic GEMM tile choice against forced 128 x 128 tiles (faer_hip_set_gemm_variant(1)) -- the V^T A
products of the block applications are short-wide outputs with a deep K, where the automatic rule picks 64 x 64 tiles + split-K"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
F = ge.load_package(); torch.cuda.set_device(0); F.lib(); F.use_torch_stream()
lib = F.lib()
for n in [int(x) for x in sys.argv[1:]] or [4096, 8192]:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    bs = int(F.qr_recommended_block_size(n, n, "float64"))
    for variant in (0, 6, 0, 6):
        lib.faer_hip_set_gemm_variant(variant)
        best = 1e9
        for rep in range(3):
            w = a.clone(); h = torch.zeros((n, bs), dtype=torch.float64, device="cuda").t()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            F.qr_factor_in_place(w, h)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"qr {n} fp64 gemm variant {variant}: {best * 1e3:.2f} ms", flush=True)
    lib.faer_hip_set_gemm_variant(0)
