"""Randomised differential test of `ICholT` (ilupp.icholt as ILU++ defines it) against oracle.icholt: random SPD systems (banded or
scrambled, 2-16 entries a row), random (add_fill_in, threshold); the device factor must equal the restatement bit for bit -- with the
candidates of a column in registers (default) and with every column through the LDS hash table (DPCG_ICHOLT_REGS=0) -- and both must
refuse the same inputs (a row or column of more than 64 kept entries, more than 256 candidates, a non-positive pivot).

    python tools/fuzz_icholt.py [cases] [seed]
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

import deeppreconditioning_amd as D
from deeppreconditioning_amd._lib import DpcgError
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def random_spd(n, per_row, band, scramble, dominance):
    k = max(1, per_row // 2)
    rows = np.repeat(np.arange(n), k)
    cols = rows - rng.integers(1, max(2, band), size=n * k)
    keep = cols >= 0
    B = sp.coo_matrix((rng.uniform(-1, 1, size=keep.sum()), (rows[keep], cols[keep])), shape=(n, n)).tocsr()
    B.sum_duplicates()
    A = B + B.T
    A = (A + sp.diags(dominance * np.asarray(abs(A).sum(axis=1)).ravel() + rng.uniform(0.1, 1.0, n))).tocsr()
    if scramble:
        q = rng.permutation(n)
        A = A[q][:, q].tocsr()
    A.sort_indices()
    return A


bad = 0
for case in range(cases):
    n = int(rng.choice([40, 300, 1000, 2400, 5000]))
    per_row = int(rng.choice([2, 4, 6, 10, 16]))
    band = int(rng.choice([8, 60, 200, n]))
    scramble = bool(rng.integers(0, 2))
    dominance = float(rng.choice([1.0, 1.0, 0.7]))            # 0.7: not diagonally dominant -- pivots may fail
    fill = int(rng.choice([0, 1, 1, 3, 10]))
    thr = float(rng.choice([0.0, 1e-4, 1e-2, 0.1, 0.3]))
    A = random_spd(n, per_row, band, scramble, dominance)
    try:
        Lref = O.icholt(A, fill, thr)
    except ValueError as e:
        Lref = str(e)
    for lds, regs in (("1", "1"), ("2", "1"), ("0", "1"), ("0", "0")):
        os.environ["DPCG_ICHOLT_LDS"] = lds
        os.environ["DPCG_ICHOLT_WAVES"] = str(int(rng.choice([4, 8, 16])))
        os.environ["DPCG_ICHOLT_REGS"] = regs
        S = D.CsrSystem.from_any(A, reorder=None)
        try:
            S.set_preconditioner(D.ICholT("multiply", add_fill_in=fill, threshold=thr))
            got = S.factor()
        except DpcgError as e:
            got = str(e)
        S.close()
        if isinstance(Lref, str) or isinstance(got, str):
            ok = isinstance(Lref, str) and isinstance(got, str)
        else:
            ok = np.array_equal(got[0], Lref.indptr) and np.array_equal(got[1], Lref.indices) and np.array_equal(got[2], Lref.data)
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} lds {lds} regs {regs}: n {n} per_row {per_row} band {band} scramble {scramble} dominance {dominance} fill {fill} thr {thr}: "
                  f"oracle {Lref if isinstance(Lref, str) else 'factor'} / device {got if isinstance(got, str) else 'factor'}", flush=True)
    what = Lref if isinstance(Lref, str) else f"nnz(L) {Lref.nnz}"
    print(f"case {case}: n {n} per_row {per_row} band {band} scramble {scramble} fill {fill} thr {thr}: {what}", flush=True)
print(f"fuzz_icholt: {cases} cases, {bad} mismatches")
