"""The multicolour ordering under DPCG_COLOR_REGIONS=0 / 1 (separate processes): a digest of the permutation and the setup time.
The two must agree vertex by vertex on bipartite graphs (same normalisation).   python tools/color_probe.py"""
import hashlib, time
import numpy as np, scipy.sparse as sp, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from oracle import oracle as O

def two_blocks():
    A = O.poisson2d(200)
    B = O.poisson3d(30)
    return sp.block_diag([A, B, A[:5000][:, :5000]]).tocsr()

cases = [("poisson2d_256", lambda: poisson.poisson_system(2, 256)), ("poisson3d_64", lambda: poisson.poisson_system(3, 64)),
         ("poisson3d_100", lambda: poisson.poisson_system(3, 100)),
         ("scrambled3d_100", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 100, 0))),
         ("scrambled2d_256_norcm", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(2, 256, 0), reorder=None)),
         ("three_components", lambda: D.CsrSystem.from_any(two_blocks(), reorder=None)),
         ("poisson2d_1024", lambda: poisson.poisson_system(2, 1024))]
for name, make in cases:
    s = make()
    s.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    nc, q = s.precond_ordering()
    torch.cuda.synchronize()
    s.set_preconditioner(D.Jacobi())
    import os
    os.environ["DPCG_KEEP_COLORING"] = "0"
    t0 = time.perf_counter()
    s.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"{name:24s} colours {nc}  perm {hashlib.sha1(q.tobytes()).hexdigest()[:12]}  setup {ms:7.2f} ms", flush=True)
    s.close()
