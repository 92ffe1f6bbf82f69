"""Scratch: lower vs upper solve time on the scrambled 1M-DoF IC(0) factor, original and index-reversed numbering."""
import sys, time
import numpy as np, scipy.sparse as sp, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

A = sp.csr_matrix(poisson.unstructured_like_csr(3, 100, 0))
n = A.shape[0]
rev = np.arange(n)[::-1]
for name, M in (("original", A), ("reversed", A[rev][:, rev].tocsr())):
    s = D.CsrSystem.from_any(M)
    s.set_preconditioner(D.IC0("solve"))
    r = poisson.rhs(n, 0)
    for upper in (False, True):
        s.sptrsv(r, upper)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            s.sptrsv(r, upper)
        torch.cuda.synchronize()
        print(f"{name} upper={upper}: {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us", flush=True)
    s.close()
