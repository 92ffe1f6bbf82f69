"""Randomised differential run of the whole-chip solve with the triangular solves inside (dpcg_chip_trsv.hip) against the C oracle
(sequential substitution + the whole-chip reduction tree): grid-like and mesh systems of 17 K .. 520 K rows in random numberings, IC(0) in
multicolour order and in the caller's order (level limit lifted), resident and streamed forms, random start vectors and caps.
    python tools/fuzz_chip_trsv.py [cases] [seed]"""
import os
import pathlib
import sys

import numpy as np
import torch

os.environ.setdefault("DPCG_CHIP_TRSV_MAX_LEVELS", "40")
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731


def permuted(A, perm):
    B = A[perm][:, perm].tocsr()
    B.sort_indices()
    return B


bad = 0
taken = 0
for case in range(cases):
    kind = rng.choice(["u3d", "p3d", "p2d", "u2d", "quad"])
    if kind == "u3d":
        g = int(rng.integers(27, 81)); A = O.unstructured_like(O.poisson3d(g), seed=int(rng.integers(0, 99))); tag = f"unstructured3d_{g}"
    elif kind == "p3d":
        g = int(rng.integers(27, 70)); A = O.poisson3d(g); tag = f"poisson3d_{g}"
    elif kind == "p2d":
        g = int(rng.integers(140, 720)); A = O.poisson2d(g); tag = f"poisson2d_{g}"
    elif kind == "u2d":
        g = int(rng.integers(140, 600)); A = O.unstructured_like(O.poisson2d(g), seed=int(rng.integers(0, 99))); tag = f"unstructured2d_{g}"
    else:
        g = int(rng.integers(150, 560)); A = O.quadtree_fv_laplacian(g, int(rng.integers(0, 9)), numbering="random"); tag = f"quadtree_random_{g}"
    n = A.shape[0]
    ordering = "multicolor" if (kind in ("p3d", "p2d") or rng.random() < 0.6) else "caller"
    resident = rng.random() < 0.75
    os.environ["DPCG_CHIP_TRSV_RESIDENT"] = "1" if resident else "0"
    S = D.CsrSystem.from_any(A)
    perm = S.permutation() if S.reordered else None
    B = permuted(A, perm) if perm is not None else A
    b = O.rhs(n, int(rng.integers(0, 1000)))
    S.set_preconditioner(D.IC0("solve", ordering=ordering) if ordering == "multicolor" else D.IC0("solve"))
    ci = S.chip_info()
    if not ci["chip_by_default"]:
        print(f"[{case}] {tag} n {n} {ordering}: not taken (levels {S.info()['levels_lower']})", flush=True)
        S.close()
        continue
    taken += 1
    if ordering == "multicolor":
        nc, q = S.precond_ordering()
        Lf = CO.ic0(permuted(A, q))
        qinv = np.empty(n, dtype=np.int32); qinv[q] = np.arange(n, dtype=np.int32)
        kw = dict(precond_perm=qinv[perm] if perm is not None else qinv)
    else:
        Lf = CO.ic0(A)
        kw = dict(precond_perm=perm) if perm is not None else {}
    tree = {**S.reduction_geometry(), "form": "chip", "rows_per_workgroup": ci["rows_per_workgroup"]}
    use_x0 = rng.random() < 0.4
    x0 = O.rhs(n, int(rng.integers(0, 1000))) if use_x0 else None
    cap = int(rng.choice([1024, 1024, 40, 3]))
    res = S.solve(dev(b), x0=dev(x0) if use_x0 else None, max_iter=cap)
    _, it, hist, x = CO.pcg(B, b[perm] if perm is not None else b, "llt_solve", L=Lf, max_iter=cap,
                            x0=(x0[perm] if perm is not None else x0) if use_x0 else None, device_tree=tree, **kw)
    xs = res.x.cpu().numpy()
    ok = res.iterations == it and np.array_equal(res.res_history, hist) and np.array_equal(xs[perm] if perm is not None else xs, x)
    bad += 0 if ok else 1
    print(f"[{case}] {tag} n {n} {ordering} {'resident' if resident else 'streamed'} levels {S.info()['levels_lower']} x0 {use_x0} cap {cap}: "
          f"{res.iterations} updates, {'EQUAL' if ok else 'MISMATCH'}", flush=True)
    S.close()
print(f"{cases} cases, {taken} taken by the one-launch form, {bad} mismatches")
sys.exit(1 if bad else 0)
