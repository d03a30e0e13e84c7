import sys; sys.path.insert(0,"tests"); sys.path.insert(0,".")
import numpy as np
from gpu_util import init_gpu, to_dev, to_host
F=init_gpu()
for dtype in (np.float32, np.float64):
  for n in (1500, 1000, 2100):
    rng=np.random.default_rng(n); a=rng.standard_normal((n,n)); a=np.asarray(a+a.T,dtype=dtype,order="F")
    ev=np.linalg.eigvalsh(a.astype(np.float64))
    for f in (0,1,0,1):
        F.lib().faer_hip_debug_level2_force_memory_bodies(f)
        vd,hd=to_dev(a),to_dev(np.zeros((8,n-1),dtype=dtype,order="F")); F.tridiag_in_place(vd,hd); v=np.array(to_host(vd)).astype(np.float64)
        from scipy.linalg import eigvalsh_tridiagonal
        d,e=np.diag(v).copy(),np.diag(v,-1).copy()
        ok=np.isfinite(d).all() and np.isfinite(e).all()
        err=np.abs(eigvalsh_tridiagonal(d,e)-ev).max() if ok else float("nan")
        print(np.dtype(dtype).name, n, "force_mem",f,"spectrum err",err)
F.lib().faer_hip_debug_level2_force_memory_bodies(0)
