import numpy as np, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from deeppreconditioning_amd.operators import release_cached_memory
from deeppreconditioning_amd.batch import solve_batch
from oracle import oracle as O
# (a) update_values + team batch
A0 = O.poisson2d(128); n = A0.shape[0]
d = np.random.default_rng(1).uniform(0.5, 2.0, n)
import scipy.sparse as sp
A1 = (sp.diags(d) @ A0 @ sp.diags(d)).tocsr(); A1.sort_indices()
S = [D.CsrSystem.from_any(A0, reorder=None) for _ in range(4)]
for s in S: s.set_preconditioner(D.Jacobi())
b = [torch.from_numpy(O.rhs(n, i)).cuda() for i in range(4)]
r0 = solve_batch(S, b)
for s in S: s.update_values(A1.data); s.set_preconditioner(D.Jacobi())
release_cached_memory()
r1 = solve_batch(S, b)
F = D.CsrSystem.from_any(A1, reorder=None); F.set_preconditioner(D.Jacobi())
ref = [F.solve(bi, flags=D._lib.NO_SMALL | D._lib.NO_TEAM) for bi in b]
print("team batch after update_values:", all(r.iterations == q.iterations and torch.allclose(r.x, q.x, rtol=1e-9, atol=1e-12) for r, q in zip(r1, ref)), [r.iterations for r in r1], [r.iterations for r in r0])
# (c) release between operations on live handles
big = poisson.poisson_system(2, 700); big.set_preconditioner(D.IC0("solve"))
bb = poisson.rhs(big.n, 0); x1 = big.solve(bb).x.clone()
release_cached_memory()
big.set_preconditioner(D.IC0("solve")); release_cached_memory()
x2 = big.solve(bb).x
print("IC0 by strips, cache released in between:", torch.equal(x1, x2))
big.close(); [s.close() for s in S]; F.close()
print("cached bytes after closing:", "released" if release_cached_memory() is None else "?")
