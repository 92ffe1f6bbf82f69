"""IC(0) in solve mode on a 27-point 3-D stencil (13 lower entries per row: every row is longer than a width-3 or width-6 record,
longer than width 12 too on a scrambled numbering): setup, levels, PCG -- natural and scrambled numbering."""
import time
import numpy as np, scipy.sparse as sp, torch
import deeppreconditioning_amd as D
from oracle import oracle as O

def stencil27(m):
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], [-1, 0, 1])
    K = sp.kron(sp.kron(T, T), T).tocsr()          # 1 on all 27 neighbours (incl. self)
    A = -K
    A.setdiag(0.0)
    A = A + sp.diags(np.asarray(-A.sum(axis=1)).ravel() + 1.0)      # diagonally dominant SPD
    A = A.tocsr(); A.sort_indices()
    return A

for m in (40, 64):
    for name, A in ((f"natural {m}^3", stencil27(m)), (f"scrambled {m}^3", O.unstructured_like(stencil27(m), 1))):
        n = A.shape[0]
        S = D.CsrSystem.from_any(A)
        b = torch.from_numpy(O.rhs(n, 0)).cuda()
        S.set_preconditioner(D.Jacobi())
        rj = S.solve(b, want_history=False); rj = S.solve(b, want_history=False)
        t0 = time.perf_counter()
        S.set_preconditioner(D.IC0("solve"))
        torch.cuda.synchronize()
        setup = (time.perf_counter() - t0) * 1e3
        r = S.solve(b, want_history=False); r = S.solve(b, want_history=False)
        info = S.info()
        print(f"{name:16s} n {n:7d} nnz/row {A.nnz / n:5.1f} reordered {S.reordered}  jacobi {rj.iterations:4d} its {rj.seconds / rj.iterations * 1e6:6.1f} us/update | "
              f"ic0 levels {info['levels_lower']:5d} setup {setup:7.1f} ms {r.iterations:4d} its {r.seconds / max(r.iterations, 1) * 1e6:8.1f} us/update status {r.status}", flush=True)
        S.close()
