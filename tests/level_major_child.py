"""Child process of test_level_major_triangular_solves: DPCG_LEVEL_MAJOR=1 (read once per process) forces the level-major form of
the triangular solves (the factor solved in its own level-order numbering; paired applies without way-in passes) on factors far
smaller than the ones that choose it by themselves.  Everything must stay bit-identical to sequential substitution, in every
interleaving of standalone solves and paired applies (the `pending` invariant of Levels::lm_out).  Prints one JSON line."""
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

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
out = {}
cases = (("unstructured3d_24", O.unstructured_like(O.poisson3d(24), seed=1), None, "ic0"),        # rows of up to 6 entries: width-6 records
         ("unstructured3d_30_rcm", O.unstructured_like(O.poisson3d(30), seed=2), "rcm", "ic0"),  # reordered handle
         ("poisson2d_90", O.poisson2d(90), None, "ic0"),                                          # width-3 records, many narrow levels
         ("unstructured2d_70_ict", O.unstructured_like(O.poisson2d(70), seed=5), None, "ict"))   # rows longer than a record
for name, A, reorder, kind in cases:
    n = A.shape[0]
    S = D.CsrSystem.from_any(A, reorder=reorder)
    S.set_preconditioner(D.IC0("solve") if kind == "ic0" else D.ICT("solve", threshold=1e-3))
    rp, ci, v = S.factor()
    L = sp.csr_matrix((v, ci, rp), shape=(n, n))
    rec = {"apply": True, "lower": True, "upper": True}
    for seed in (4, 5, 6):
        b = O.rhs(n, seed)
        t = CO.sptrsv_lower(L, b)
        z = CO.sptrsv_upper(CO.transpose_csr(L), t)
        # paired, paired, standalone lower, paired, standalone upper, paired: every hand-over of the pending preset
        rec["apply"] &= bool(np.array_equal(S.precond_apply(dev(b)).cpu().numpy(), z))
        rec["apply"] &= bool(np.array_equal(S.precond_apply(dev(b)).cpu().numpy(), z))
        rec["lower"] &= bool(np.array_equal(S.sptrsv(dev(b), upper=False).cpu().numpy(), t))
        rec["apply"] &= bool(np.array_equal(S.precond_apply(dev(b)).cpu().numpy(), z))
        rec["upper"] &= bool(np.array_equal(S.sptrsv(dev(t), upper=True).cpu().numpy(), z))
        rec["apply"] &= bool(np.array_equal(S.precond_apply(dev(b)).cpu().numpy(), z))
    b = O.rhs(n, 4)
    # the oracle runs on the system the handle iterates on (P A P^T with the caller's factor applied as P M P^T on a
    # reordered handle: orc_pcg_perm), so the bar is north_star's 1e-10 in every case
    if S.reordered:
        perm = S.permutation()
        Bp = A[perm][:, perm].tocsr()
        Bp.sort_indices()
        _, it, hist, _ = CO.pcg(Bp, b[perm], "llt_solve", L=L, precond_perm=perm)
    else:
        _, it, hist, _ = CO.pcg(A, b, "llt_solve", L=L)
    # (NO_SMALL: the launches -- a plain call of the 27 000-row system would be the one-launch kernel with the triangular solves inside)
    res = S.solve(dev(b), flags=D._lib.NO_SMALL)      # first solve: graph capture comes before the first real apply
    res2 = S.solve(dev(b), flags=D._lib.NO_GRAPH)
    rec["iterations"] = [res.iterations, res2.iterations, it]
    rec["hist_rel"] = float(np.max(np.abs(res.res_history - hist) / hist)) if res.iterations == it else None
    rec["same_bits_without_graph"] = bool(np.array_equal(res.res_history, res2.res_history))
    rec["levels"] = S.info()["levels_lower"]
    out[name] = rec
    S.close()
print(json.dumps(out), flush=True)
