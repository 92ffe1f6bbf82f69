"""What would a handle numbered colour by colour buy?  The system permuted in Python by the library's own multicolour ordering
(so that the factor's numbering is the handle's: every map of the sweeps the identity), Jacobi and IC(0)-multicolour PCG on both."""
import numpy as np, scipy.sparse as sp, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from oracle import oracle as O

for name, A, kw in (("poisson3d_100", O.poisson3d(100), {}), ("scrambled3d_100", poisson.unstructured_like_csr(3, 100, 0), {}),
                    ("poisson2d_256", O.poisson2d(256), {})):
    S = D.CsrSystem.from_any(A, **kw)
    b = O.rhs(A.shape[0], 0)
    S.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    nc, q = S.precond_ordering()
    for label, sysm, rhs in (("as is", S, b), ("colour-major", None, None)):
        if sysm is None:
            A2 = A[q][:, q].tocsr(); A2.sort_indices()
            sysm = D.CsrSystem.from_any(A2, reorder=None)
            rhs = b[q]
        bd = torch.from_numpy(rhs).cuda()
        out = []
        for pc in (D.Jacobi(), D.IC0("solve", ordering="multicolor")):
            sysm.set_preconditioner(pc)
            sysm.solve(bd, want_history=False)
            r = sysm.solve(bd, want_history=False)
            out.append(f"{r.iterations} its {r.seconds * 1e3:.3f} ms = {r.seconds / r.iterations * 1e6:.1f} us/update")
        q2 = sysm.precond_ordering()[1]
        print(f"{name:16s} {label:13s} jacobi {out[0]}   ic0-mc {out[1]}   spmv {sysm.info()['spmv_kernel']}  identity ordering {bool(np.array_equal(q2, np.arange(len(q2))))}", flush=True)
