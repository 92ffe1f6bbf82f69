"""How many rows beyond the 3-entry ring / strip records does it take for the sync-free kernels (wide records) to beat those plans?
A natural-order 7-point grid with one extra lower neighbour (i-1, j-1, k) on a share of the rows; IC(0) in solve mode."""
import sys
import numpy as np, scipy.sparse as sp, torch
import deeppreconditioning_amd as D
from oracle import oracle as O

m = int(sys.argv[1]) if len(sys.argv) > 1 else 48
rng = np.random.default_rng(0)
A0 = O.poisson3d(m).tolil()
n = m ** 3
for share in (0.0, 0.05, 0.2, 0.5, 1.0):
    A = sp.lil_matrix(A0)
    rows = np.nonzero(rng.uniform(size=n) < share)[0]
    rows = rows[(rows % m != 0) & ((rows // m) % m != 0)]          # have an (i-1, j-1) neighbour in the same plane
    cols = rows - m - 1
    E = sp.coo_matrix((np.full(rows.size, -0.5), (rows, cols)), shape=(n, n)).tocsr()
    B = (sp.csr_matrix(A0) + E + E.T + sp.diags(np.asarray(abs(E + E.T).sum(axis=1)).ravel())).tocsr()
    B.sort_indices()
    S = D.CsrSystem.from_any(B, reorder=None)
    b = torch.from_numpy(O.rhs(n, 0)).cuda()
    S.set_preconditioner(D.IC0("solve"))
    r = S.solve(b, want_history=False); r = S.solve(b, want_history=False)
    print(f"{m}^3 share of 4-entry rows {share:4.2f}: levels {S.info()['levels_lower']:4d} {r.iterations:3d} its {r.seconds / max(r.iterations, 1) * 1e6:8.1f} us/update", flush=True)
    S.close()
