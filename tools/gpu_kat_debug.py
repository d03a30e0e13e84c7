"""debug: exact KATs on the GPU, where the first mismatch is"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_util import init_gpu, to_dev, to_host
from test_exact_kats import K, full
F = init_gpu()
for case in K["lu"]:
    for dtype in (np.float64, np.float32):
        for general in (0, 1):
            a = full(case["a"], dtype)
            d = to_dev(a)
            F.lib().faer_hip_debug_lu_force_general(general)
            perm, _, nt = F.partial_piv_lu_factor_in_place(d)
            F.lib().faer_hip_debug_lu_force_general(0)
            got = to_host(d); want = full(case["lu"], dtype)
            perm = perm.astype(np.int64)
            eq = (got == want) | ((got != got) & (want != want))
            pok = list(perm) == case["perm"]
            msg = f"{case['name']} {dtype.__name__} general={general}: perm ok {pok} nt {nt} vs {case['transpositions']}, entries equal {eq.all()}"
            if not eq.all() or not pok:
                bad = np.argwhere(~eq)
                cols = sorted(set(bad[:, 1]))[:10]
                pj = [i for i in range(len(perm)) if perm[i] != case["perm"][i]][:10]
                msg += f"; first bad columns {cols}, first bad entries {bad[:5].tolist()}, perm diffs at {pj}"
                if len(bad):
                    i, j = bad[0]
                    msg += f" got {got[i, j]!r} want {want[i, j]!r}"
            print(msg, flush=True)
