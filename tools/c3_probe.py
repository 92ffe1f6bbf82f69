import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
for dim, n in ((3, 16), (3, 100)):
    t = time.time(); A = poisson.unstructured_like_csr(dim, n, 0); t_gen = time.time() - t
    if n == 16:
        from oracle import oracle as O
        ref = O.unstructured_like(O.poisson3d(16), 0)
        print("matches oracle generator:", abs(A - ref).max() == 0 and np.array_equal(A.indices, ref.indices))
    S = D.CsrSystem.from_any(A)
    b = poisson.rhs(S.n, 0)
    print(n, "gen s", round(t_gen, 2), S.info())
    S.set_preconditioner(D.Jacobi()); r = S.solve(b); print(" jacobi", r.iterations, r.status, round(r.seconds * 1e3, 2), "ms", round(r.iterations / r.seconds), "it/s")
    t = time.time(); S.set_preconditioner(D.IC0("solve")); print(" ic0 setup s", round(time.time() - t, 2), S.info())
    r = S.solve(b); r = S.solve(b); print(" ic0-solve", r.iterations, r.status, round(r.seconds * 1e3, 2), "ms", round(r.iterations / r.seconds), "it/s")
    r = S.solve(b, flags=D._lib.SPMV_F32); print(" ic0-solve mixed", r.iterations, round(r.seconds * 1e3, 2))
    S.set_preconditioner(D.Jacobi()); r = S.solve(b, flags=D._lib.SPMV_F32); r = S.solve(b, flags=D._lib.SPMV_F32); print(" jacobi mixed", r.iterations, round(r.seconds * 1e3, 2), "ms", round(r.iterations / r.seconds), "it/s")
S = poisson.poisson_system(3, 100); b = poisson.rhs(S.n, 0)
S.set_preconditioner(D.IC0("solve")); print("natural 100^3 ic0", S.info()); r = S.solve(b); r = S.solve(b); print(" ic0-solve", r.iterations, round(r.seconds * 1e3, 2), "ms")
S.set_preconditioner(D.Jacobi()); r = S.solve(b, flags=D._lib.SPMV_F32); r = S.solve(b, flags=D._lib.SPMV_F32); print(" natural jacobi mixed", r.iterations, round(r.seconds * 1e3, 2), "ms", round(r.iterations / r.seconds), "it/s")
print("---- RCM reordering of the 1M-DoF scrambled system ----")
A = poisson.unstructured_like_csr(3, 100, 0)
t = time.time(); R = D.CsrSystem.from_any(A, reorder="rcm"); print("rcm setup s", round(time.time() - t, 2), R.info())
b = poisson.rhs(R.n, 0)
R.set_preconditioner(D.Jacobi()); r = R.solve(b); r = R.solve(b); print(" rcm jacobi", r.iterations, round(r.seconds * 1e3, 2), "ms", round(r.iterations / r.seconds), "it/s")
R.set_preconditioner(D.IC0("solve")); print(R.info()); r = R.solve(b); r = R.solve(b); print(" rcm ic0-solve", r.iterations, round(r.seconds * 1e3, 2), "ms", round(r.iterations / r.seconds), "it/s")
