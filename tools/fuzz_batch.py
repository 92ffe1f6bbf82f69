"""Randomised concurrency test: groups of random sparse SPD systems (sizes on both sides of the kernel-selection boundaries,
banded or scrambled, every preconditioner kind) solved one at a time and then as a `solve_batch` on 2..8 streams, repeatedly,
with a GEMM stream keeping the CUs busy: every x, count and final residual must be bit-identical to the one-at-a-time solve.

    python tools/fuzz_batch.py [groups] [seed]
"""
import sys
import numpy as np
import scipy.sparse as sp
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd.batch import solve_batch

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
SIZES = [600, 3000, 6145, 9000, 20000, 40000, 70000, 120000, 400000]
KINDS = ["none", "jacobi", "ic0_solve", "ic0_multiply", "ict_solve", "csr"]


def random_spd(n, per_row, band, scramble):
    k = max(1, per_row // 2)
    rows = np.repeat(np.arange(n), k)
    offs = rng.integers(1, max(2, band), size=n * k)
    cols = rows - offs
    keep = cols >= 0
    vals = rng.uniform(-1, 1, size=keep.sum())
    B = sp.coo_matrix((vals, (rows[keep], cols[keep])), shape=(n, n)).tocsr()
    B.sum_duplicates()
    A = B + B.T
    A = (A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + rng.uniform(0.1, 1.0, n))).tocsr()
    if scramble:
        p = rng.permutation(n)
        A = A[p][:, p].tocsr()
    A.sort_indices()
    return A


def grid(n):       # a few real grids among the random matrices (strip plans, level-major factors)
    from oracle import oracle as O
    pick = rng.integers(0, 4)
    if pick == 0:
        return O.poisson2d(int(np.sqrt(n)))
    if pick == 1:
        return O.poisson3d(max(8, int(round(n ** (1 / 3)))))
    if pick == 2:
        return O.unstructured_like(O.poisson2d(int(np.sqrt(n))), int(rng.integers(0, 100)))
    return O.unstructured_like(O.poisson3d(max(8, int(round(n ** (1 / 3))))), int(rng.integers(0, 100)))


side = torch.cuda.Stream()
ga = torch.randn(3072, 3072, device="cuda")
gb = torch.randn(3072, 3072, device="cuda")
bad = 0
for g in range(groups):
    systems, rhs, tags = [], [], []
    for _ in range(int(rng.integers(3, 8))):
        n = int(rng.choice(SIZES))
        if rng.integers(0, 2):
            A = grid(n)
        else:
            A = random_spd(n, int(rng.choice([2, 4, 6, 10])), int(rng.choice([2, 8, 64, 1000, n])), bool(rng.integers(0, 2)))
        n = A.shape[0]
        kind = str(rng.choice(KINDS if n <= 70000 else KINDS[:3]))
        S = D.CsrSystem.from_any(A, reorder=str(rng.choice(["auto", "auto", "rcm"])))
        try:
            if kind == "csr":
                L = sp.tril(A, format="csr")
                S.set_preconditioner(D.CsrPreconditioner(sp.diags(1.0 / A.diagonal()).tocsr() if rng.integers(0, 2) else (L @ L.T).tocsr()))
            else:
                S.set_preconditioner({"none": None, "jacobi": D.Jacobi(), "ic0_solve": D.IC0("solve"), "ic0_multiply": D.IC0("multiply"),
                                      "ict_solve": D.ICT("solve", 1, 0.01)}.get(kind))
        except D._lib.DpcgError as e:      # a random matrix IC breaks down on: nothing to compare
            S.close()
            continue
        systems.append(S)
        rhs.append(torch.from_numpy(rng.uniform(-1, 1, n)).cuda())
        tags.append(f"n={n} {kind} reordered={S.reordered}")
    if not systems:
        continue
    max_iter = int(rng.choice([40, 200, 1024]))
    # a mixed batch runs every system through the general (multi-kernel) path, a one-at-a-time solve of a small system would
    # take the whole-solve kernel (other reduction orders: equal to ~1e-15, not to the bit): compare like with like
    # (the same for the team kernel of mid-size systems: forced for both runs when every system of the group is eligible,
    # forbidden for both otherwise)
    flags = int(rng.choice([0, 0, D._lib.NO_FUSE, D._lib.NO_GRAPH])) | D._lib.NO_SMALL
    team_ok = all(6144 < S.n <= 65536 and not S.reordered and S.info()["precond"] in (0, 1) and S.info()["spmv_kernel"] != "vector"
                  for S in systems) and not (flags & D._lib.NO_FUSE)
    flags |= D._lib.TEAM if (team_ok and rng.integers(0, 2)) else D._lib.NO_TEAM
    single = [S.solve(b, max_iter=max_iter, flags=flags, want_history=False) for S, b in zip(systems, rhs)]
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(8):
                gb = torch.mm(ga, gb).mul_(1e-3)
        out = solve_batch(systems, rhs, max_iter=max_iter, flags=flags, n_streams=int(rng.choice([2, 3, 4, 8])))
        for s1, o, tag in zip(single, out, tags):
            same = o.iterations == s1.iterations and o.status == s1.status and (torch.equal(o.x, s1.x) or
                                                                               (s1.status == 2 and bool(torch.isnan(s1.x).any())))
            if not same:
                bad += 1
                d = (o.x - s1.x).abs()
                print(f"MISMATCH group {g} rep {rep} flags {flags} max_iter {max_iter}: {tag}: its {o.iterations}/{s1.iterations} status "
                      f"{o.status}/{s1.status} differing {int((d > 0).sum())} of {d.numel()} max rel {float(d.max() / s1.x.abs().max()):.2e} "
                      f"final_res {o.final_res:.17e} / {s1.final_res:.17e}", flush=True)
    side.synchronize()
    for S in systems:
        S.close()
    print(f"group {g}: {len(systems)} systems, flags {flags}, max_iter {max_iter}: ok so far ({bad} mismatches)", flush=True)
print(f"fuzz_batch: {groups} groups, {bad} mismatches")
