"""IC(0) in solve mode on the 9-point 2-D stencil (4 lower entries per row: one more than the ring / strip records hold)."""
import time
import numpy as np, scipy.sparse as sp, torch
import deeppreconditioning_amd as D
from oracle import oracle as O

def stencil9(m):
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], [-1, 0, 1])
    A = -sp.kron(T, T).tocsr()
    A.setdiag(0.0)
    A = (A + sp.diags(np.asarray(-A.sum(axis=1)).ravel() + 0.05)).tocsr()
    A.sort_indices()
    return A

for m in (256, 512):
    A = stencil9(m)
    n = A.shape[0]
    S = D.CsrSystem.from_any(A)
    b = torch.from_numpy(O.rhs(n, 0)).cuda()
    t0 = time.perf_counter(); S.set_preconditioner(D.IC0("solve")); torch.cuda.synchronize()
    setup = (time.perf_counter() - t0) * 1e3
    r = S.solve(b, want_history=False); r = S.solve(b, want_history=False)
    print(f"9-point {m}^2 n {n:7d} levels {S.info()['levels_lower']:5d} setup {setup:7.1f} ms {r.iterations:4d} its {r.seconds / max(r.iterations, 1) * 1e6:8.1f} us/update status {r.status}", flush=True)
    S.close()
