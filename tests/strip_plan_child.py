"""Child process of test_strip_pipelined_triangular_solves: with DPCG_SETUP_TRACE=1 (read once per process) the library reports
on stderr whether the strip plan of a factor was kept; the triangular solves must be bit-identical to sequential substitution
either way.  Prints one JSON line."""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

import deeppreconditioning_amd as D  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import oracle as O  # noqa: E402

out = {}
for name, A in (("poisson3d_40", O.poisson3d(40)), ("poisson2d_300", O.poisson2d(300)),
                ("unstructured3d_36_rcm", O.unstructured_like(O.poisson3d(36), seed=3)),
                ("poisson3d_56", O.poisson3d(56)), ("poisson2d_400", O.poisson2d(400))):      # large enough for the two-way cut
    n = A.shape[0]
    S = D.CsrSystem.from_any(A, reorder="rcm" if name.endswith("rcm") else None)
    L = CO.ic0(A)
    b = O.rhs(n, 4)
    ok = {}
    for mode, pc in (("ic0", D.IC0("solve")), ("user_factor", D.LLtSolve(L))):
        S.set_preconditioner(pc)
        t = CO.sptrsv_lower(L, b)
        z = CO.sptrsv_upper(CO.transpose_csr(L), t)
        lo = S.sptrsv(torch.from_numpy(b).cuda(), upper=False).cpu().numpy()
        up = S.sptrsv(torch.from_numpy(t).cuda(), upper=True).cpu().numpy()
        ap = S.precond_apply(torch.from_numpy(b).cuda()).cpu().numpy()
        res = S.solve(torch.from_numpy(b).cuda())
        if S.reordered:      # oracle on P A P^T with the caller's factor applied as P M P^T (orc_pcg_perm): 1e-10 applies
            perm = S.permutation()
            Bp = A[perm][:, perm].tocsr()
            Bp.sort_indices()
            _, it, hist, _ = CO.pcg(Bp, b[perm], "llt_solve", L=L, precond_perm=perm)
        else:
            _, it, hist, _ = CO.pcg(A, b, "llt_solve", L=L)
        ok[mode] = {"lower": bool(np.array_equal(lo, t)), "upper": bool(np.array_equal(up, z)), "apply": bool(np.array_equal(ap, z)),
                    "iterations": [res.iterations, it],
                    "hist_rel": float(np.max(np.abs(res.res_history - hist) / hist)) if res.iterations == it else None}
    out[name] = {"levels": S.info()["levels_lower"], **ok}
    S.close()
print(json.dumps(out), flush=True)
