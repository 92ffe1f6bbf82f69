"""Level widths of IC(0) (caller's order) on the config-3 stand-in and how the solve is segmented; time per apply."""
import sys, pathlib, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from oracle import oracle as O, c_oracle as CO
A = poisson.unstructured_like_csr(3, 100, 0)
S = D.CsrSystem.from_any(A)
S.set_preconditioner(D.IC0("solve"))
print(S.info())
rp, ci, v = S.factor()
import scipy.sparse as sp
n = S.n
L = sp.csr_matrix((v, ci, rp), shape=(n, n))
# levels of the lower solve
lvl = np.zeros(n, dtype=np.int32)
indptr, indices = L.indptr, L.indices
for i in range(n):
    s, e = indptr[i], indptr[i + 1] - 1
    if e > s:
        lvl[i] = lvl[indices[s:e]].max() + 1
print("levels", lvl.max() + 1, "widths", np.bincount(lvl).tolist())
b = torch.from_numpy(O.rhs(n, 0)).cuda()
S.precond_apply(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    S.precond_apply(b)
torch.cuda.synchronize()
print("apply us", (time.perf_counter() - t0) / 50 * 1e6)
