"""The whole-chip solve with the triangular solves inside (dpcg_chip_trsv.hip) against the C oracle (form "chip", kind llt_solve) bit for bit,
and against the launches in time:  python tools/chip_trsv_probe.py [case ...]   (cases: u60 u80 u100 p2d256 p3d41 q400)"""
import os
import pathlib
import sys
import time

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    "p3d41": lambda: O.poisson3d(41),
    "p2d256": lambda: O.poisson2d(256),
    "u60": lambda: O.unstructured_like(O.poisson3d(60), seed=1),
    "u80": lambda: O.unstructured_like(O.poisson3d(80), seed=0),
    "u100": lambda: O.unstructured_like(O.poisson3d(100), seed=0),
    "q400": lambda: O.quadtree_fv_laplacian(400, 5, numbering="random"),
}
CHECK = os.environ.get("PROBE_CHECK", "1") == "1"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def permuted(A, perm):
    B = A[perm][:, perm].tocsr()
    B.sort_indices()
    return B


def run(name):
    A = CASES[name]()
    n = A.shape[0]
    b = O.rhs(n, 0)
    S = D.CsrSystem.from_any(A)
    perm = S.permutation() if S.reordered else np.arange(n)
    B = permuted(A, perm) if S.reordered else A
    S.set_preconditioner(D.Jacobi())
    rj = S.solve(dev(b), want_history=False)
    rj = S.solve(dev(b), want_history=False)
    print(f"{name}: n {n} reordered {S.reordered}; jacobi {rj.iterations} updates, {rj.seconds / max(rj.iterations, 1) * 1e6:.2f} us per update, {rj.seconds * 1e3:.2f} ms", flush=True)
    for ordering in ("multicolor", "caller"):
        if ordering == "caller" and name in ("p3d41", "p2d256"):
            continue                                           # natural order of a grid: hundreds of levels, the launches keep it
        t0 = time.perf_counter()
        S.set_preconditioner(D.IC0("solve", ordering=ordering) if ordering == "multicolor" else D.IC0("solve"))
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t0
        t0 = time.perf_counter()
        ci = S.chip_info()
        torch.cuda.synchronize()
        t_lists = time.perf_counter() - t0
        info = S.info()
        print(f"  IC(0) {ordering}: levels {info['levels_lower']}/{info['levels_upper']}, chip_by_default {ci['chip_by_default']}, setup {t_setup * 1e3:.1f} ms, lists {t_lists * 1e3:.1f} ms", flush=True)
        res = S.solve(dev(b))
        res = S.solve(dev(b))
        multi = S.solve(dev(b), flags=D._lib.NO_SMALL)
        multi = S.solve(dev(b), flags=D._lib.NO_SMALL)
        print(f"    chip: {res.iterations} updates status {res.status}, {res.seconds / max(res.iterations, 1) * 1e6:.2f} us per update, {res.seconds * 1e3:.2f} ms | "
              f"launches: {multi.iterations} updates, {multi.seconds / max(multi.iterations, 1) * 1e6:.2f} us per update, {multi.seconds * 1e3:.2f} ms", flush=True)
        if not CHECK:
            continue
        tree = {**S.reduction_geometry(), "form": "chip", "rows_per_workgroup": ci["rows_per_workgroup"]}
        if ordering == "multicolor":
            nc, q = S.precond_ordering()
            Lf = CO.ic0(permuted(A, q))
            qinv = np.empty(n, dtype=np.int32)
            qinv[q] = np.arange(n, dtype=np.int32)
            kw = dict(precond_perm=qinv[perm])
        else:
            Lf = CO.ic0(A)
            kw = dict(precond_perm=perm) if S.reordered else {}
        _, it, hist, x = CO.pcg(B, b[perm], "llt_solve", L=Lf, device_tree=tree, **kw)
        same = res.iterations == it and np.array_equal(res.res_history, hist)
        xs = res.x.cpu().numpy()[perm]
        first = int(np.argmax(res.res_history[:min(len(hist), len(res.res_history))] != hist[:min(len(hist), len(res.res_history))])) if not same else -1
        print(f"    oracle: {it} updates; history bit for bit: {same} (first difference at {first}); x bit for bit: {np.array_equal(xs, x)}; "
              f"vs launches: count {multi.iterations == res.iterations}, hist rel {np.max(np.abs(multi.res_history[:20] - res.res_history[:20]) / res.res_history[:20]):.2e}", flush=True)
        x0 = O.rhs(n, 5)
        r0 = S.solve(dev(b), x0=dev(x0), max_iter=25)
        _, it0, hist0, _ = CO.pcg(B, b[perm], "llt_solve", L=Lf, x0=x0[perm], max_iter=25, device_tree=tree, **kw)
        print(f"    with x0: {r0.iterations == it0 and np.array_equal(r0.res_history, hist0)}", flush=True)
    S.close()


for c in (sys.argv[1:] or ["u60", "p3d41", "p2d256", "u80", "q400", "u100"]):
    run(c)
