"""One config-2 solve with a learned-like factor applied as z = L (L^T r) (15 entries a row), for rocprofv3 --kernel-trace."""
import numpy as np, torch
import deeppreconditioning_amd as D
from oracle import oracle as O
A = O.poisson2d(256)
L = O.learned_like_factor_preconditioning(A, seed=1)
S = D.CsrSystem.from_any(A)
S.set_preconditioner(D.LLtMultiply(L))
b = torch.from_numpy(O.rhs(A.shape[0], 0)).cuda()
S.solve(b, want_history=False)
r = S.solve(b, want_history=False)
print(r.iterations, r.seconds * 1e3, "ms", r.seconds / r.iterations * 1e6, "us/update")
