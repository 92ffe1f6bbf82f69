"""Randomised check of IC(0) / ICT through the schedule (strip plan / LDS-ring walk) against the sequential restatement: grid-like
patterns -- 2-D / 3-D boxes of random extents, random positive coefficients, a share of the edges removed, a third of the
2-D ones with a diagonal neighbour (triangles: cross terms), natural or scrambled numbering (the latter through the library's
reordering) -- from a few thousand to ~700K rows; a share of the smaller ones with ICT(1, 0 / 0.02 / 0.1) instead of IC(0).
The factor must equal oracle/pcg_oracle.c's bit for bit, and so must both triangular solves on the schedule that was kept.

    python tools/fuzz_ic0_setup.py [cases] [seed]
"""
import sys
import numpy as np
import scipy.sparse as sp
import torch
import deeppreconditioning_amd as D
from oracle import c_oracle as CO

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def grid_matrix(shape, drop, diagonal=False):
    """SPD matrix on a box grid: random negative couplings between grid neighbours (a share `drop` removed), dominant diagonal.
    diagonal (2-D): the south-west / north-east neighbours too -- a pattern with triangles, i.e. cross terms in IC(0)."""
    n = int(np.prod(shape))
    idx = np.arange(n).reshape(shape)
    rows, cols = [], []
    for ax in range(len(shape)):
        a = np.take(idx, np.arange(shape[ax] - 1), axis=ax).ravel()
        b = np.take(idx, np.arange(1, shape[ax]), axis=ax).ravel()
        keep = rng.uniform(size=a.size) >= drop
        rows.append(a[keep])
        cols.append(b[keep])
    if diagonal and len(shape) == 2:
        a, b = idx[:-1, :-1].ravel(), idx[1:, 1:].ravel()
        keep = rng.uniform(size=a.size) >= drop
        rows.append(a[keep])
        cols.append(b[keep])
    r, c = np.concatenate(rows), np.concatenate(cols)
    w = -rng.uniform(0.2, 2.0, r.size)
    off = sp.coo_matrix((w, (r, c)), shape=(n, n)).tocsr()
    off = off + off.T
    A = off + sp.diags(np.asarray(abs(off).sum(axis=1)).ravel() + rng.uniform(0.05, 0.5, n))
    A = A.tocsr()
    A.sort_indices()
    return A


bad = 0
for case in range(cases):
    dim = int(rng.choice([2, 2, 3]))
    target = int(rng.choice([5000, 20000, 66000, 130000, 140000, 300000, 700000]))
    if dim == 2:
        nx = int(rng.integers(max(8, int(target ** 0.5 / 3)), int(target ** 0.5 * 2)))
        shape = (max(2, target // nx), nx)
    else:
        side = max(4, int(round(target ** (1 / 3))))
        shape = (max(2, target // (side * side)), int(side * rng.uniform(0.7, 1.3)) + 1, side)
    drop = float(rng.choice([0.0, 0.0, 0.05, 0.3]))
    diagonal = dim == 2 and bool(rng.integers(0, 3) == 0)
    A = grid_matrix(shape, drop, diagonal)
    n = A.shape[0]
    scramble = bool(rng.integers(0, 3) == 0)
    if scramble:
        p = rng.permutation(n)
        A = A[p][:, p].tocsr()
        A.sort_indices()
    reorder = str(rng.choice(["auto", None, "rcm"])) if scramble else None
    reorder = None if reorder == "None" else reorder
    mode = str(rng.choice(["solve", "solve", "multiply"]))
    # a share of the smaller cases run ICT (level-1 fill, drop rule) against the Python restatement instead of IC(0)
    ict_thr = float(rng.choice([0.0, 0.02, 0.1])) if ((n <= 70000 and rng.integers(0, 2) == 0) or (n <= 310000 and rng.integers(0, 4) == 0)) else None
    tag = (f"case {case}: shape={shape} n={n} drop={drop} diagonal={diagonal} scramble={scramble} reorder={reorder} mode={mode} "
           f"{'ict thr ' + str(ict_thr) if ict_thr is not None else 'ic0'}")
    S = D.CsrSystem.from_any(A, reorder=reorder)
    if ict_thr is None:
        S.set_preconditioner(D.IC0(mode))
        Lref = CO.ic0(A)
    else:
        from oracle import oracle as O
        S.set_preconditioner(D.ICT(mode, 1, ict_thr))
        Lref = O.ict(A, 1, ict_thr)
    rp, ci, v = S.factor()
    ok = np.array_equal(rp, Lref.indptr) and np.array_equal(ci, Lref.indices) and np.array_equal(v, Lref.data)
    if ok and mode == "solve":
        r = rng.uniform(-1, 1, n)
        y_ref = CO.sptrsv_lower(Lref, r)
        z_ref = CO.sptrsv_upper(CO.transpose_csr(Lref), y_ref)
        ok = (np.array_equal(S.sptrsv(torch.from_numpy(r).cuda(), upper=False).cpu().numpy(), y_ref)
              and np.array_equal(S.sptrsv(torch.from_numpy(y_ref).cuda(), upper=True).cpu().numpy(), z_ref)
              and np.array_equal(S.precond_apply(torch.from_numpy(r).cuda()).cpu().numpy(), z_ref))
        lv = S.info()["levels_lower"]
    else:
        lv = -1
    if not ok:
        bad += 1
        print("IC0 SETUP MISMATCH", tag, flush=True)
    elif len(sys.argv) > 3:
        print("ok", tag, "levels", lv, flush=True)
    S.close()
print(f"fuzz_ic0_setup: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
