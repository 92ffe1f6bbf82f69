"""dpcg_reorder on the 1M-DoF unstructured stand-in: time of create (upload + measurement + RCM + permute + plan), the
plan it ends with, bandwidth before / after, Jacobi PCG rate."""
import time

import numpy as np
import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

for dim, n in ((3, 100), (2, 1024), (3, 64)):
    A = poisson.unstructured_like_csr(dim, n, 0)
    b = poisson.rhs(A.shape[0], 0)
    for mode in (None, "rcm", "auto"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S = D.CsrSystem.from_any(A, reorder=mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        info = S.info()
        bw = -1
        if S.reordered:
            perm = S.permutation()
            B = A[perm][:, perm].tocoo()
            bw = int(np.abs(B.row - B.col).max())
        S.set_preconditioner(D.Jacobi())
        S.solve(b, want_history=False)
        r = S.solve(b, want_history=False)
        ms = S.spmv_dot_bench(100)
        print(f"scrambled{dim}d_{n} reorder={mode!s:5s}: create {dt * 1e3:8.1f} ms  reordered {info['reordered']}  ratio {info['gather_ratio']:.2f} "
              f"kernel {info['spmv_kernel']:6s} bandwidth {bw:8d}  jacobi {r.iterations} its {r.iterations / r.seconds:9.1f} it/s  "
              f"loop-spmv {ms * 1e3:7.2f} us", flush=True)
        S.close()
