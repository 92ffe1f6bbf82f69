"""Odd inputs for IC(0) in multicolour order: a diagonal matrix, isolated vertices, a tiny system, two components of very different
size, a star.  Each: the factor against the oracle on Q A Q^T, one apply against sequential substitution, a solve."""
import numpy as np, scipy.sparse as sp, torch
import deeppreconditioning_amd as D
from oracle import oracle as O, c_oracle as CO

def check(name, A):
    A = A.tocsr(); A.sort_indices()
    n = A.shape[0]
    S = D.CsrSystem.from_any(A, reorder=None)
    S.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    nc, q = S.precond_ordering()
    B = A[q][:, q].tocsr(); B.sort_indices()
    L = CO.ic0(B)
    rp, ci, v = S.factor()
    ok = np.array_equal(rp, L.indptr) and np.array_equal(ci, L.indices) and np.array_equal(v, L.data)
    b = O.rhs(n, 1)
    z = np.empty(n); z[q] = CO.sptrsv_upper(CO.transpose_csr(L), CO.sptrsv_lower(L, b[q]))
    ok = ok and np.array_equal(S.precond_apply(torch.from_numpy(b).cuda()).cpu().numpy(), z)
    r = S.solve(torch.from_numpy(b).cuda())
    S.update_values(A.data); S.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    ok = ok and np.array_equal(S.precond_apply(torch.from_numpy(b).cuda()).cpu().numpy(), z)
    print(f"{name:28s} n {n:7d} colours {nc} levels {S.info()['levels_lower']} iterations {r.iterations} status {r.status}  {'ok' if ok else 'MISMATCH'}", flush=True)
    S.close()
    return ok

good = True
good &= check("diagonal", sp.diags(np.linspace(1.0, 2.0, 50000)))
P = O.poisson2d(200).tolil(); P[5, :] = 0; P[:, 5] = 0; P[5, 5] = 3.0; P[777, :] = 0; P[:, 777] = 0; P[777, 777] = 2.0
good &= check("isolated vertices", P.tocsr())
good &= check("tiny", O.poisson2d(6))
good &= check("small whole-solve size", O.poisson2d(38))
good &= check("two components", sp.block_diag([O.poisson3d(40), O.poisson2d(5)]))
n = 40000
star = sp.coo_matrix((np.full(n - 1, -0.001), (np.zeros(n - 1, dtype=int), np.arange(1, n))), shape=(n, n)).tocsr()
good &= check("star (one row of 40K entries)", star + star.T + sp.diags(np.full(n, 50.0)))
print("all ok" if good else "FAILURES")
