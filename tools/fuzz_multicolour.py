"""Randomised check of IC(0) in multicolour order through the colour-sweep kernels (tiled sweeps, the last lower level opening the
upper solve, the first riding on the r update): grid-like patterns as tools/fuzz_ic0_setup.py's -- 2-D / 3-D boxes of random
extents, random coefficients, a share of the edges removed, a third of the 2-D ones with a diagonal neighbour (triangles: three
or more colours), natural or scrambled numbering (the latter through the library's reordering or not) -- mostly beyond 262 144
rows, where the sweeps are chosen.  With Q from precond_ordering(): the factor must equal oracle IC(0) of Q A Q^T bit for bit, an
apply sequential substitution bit for bit (twice), and a PCG solve oracle/pcg_oracle.c's (count equal, history 1e-10).

    python tools/fuzz_multicolour.py [cases] [seed] [verbose]
"""
import sys
import numpy as np
import scipy.sparse as sp
import torch
import deeppreconditioning_amd as D
from oracle import c_oracle as CO

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def grid_matrix(shape, drop, diagonal=False):
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
    A = (off + sp.diags(np.asarray(abs(off).sum(axis=1)).ravel() + rng.uniform(0.05, 0.5, n))).tocsr()
    A.sort_indices()
    return A


bad = 0
bitwise = 0
for case in range(cases):
    dim = int(rng.choice([2, 3, 3]))
    target = int(rng.choice([60000, 270000, 300000, 420000, 600000, 1000000]))
    if dim == 2:
        nx = int(rng.integers(max(8, int(target ** 0.5 / 2)), int(target ** 0.5 * 2)))
        shape = (max(2, target // nx), nx)
    else:
        side = max(4, int(round(target ** (1 / 3))))
        shape = (max(2, target // (side * side)), int(side * rng.uniform(0.7, 1.3)) + 1, side)
    drop = float(rng.choice([0.0, 0.0, 0.05, 0.3]))
    diagonal = dim == 2 and bool(rng.integers(0, 3) == 0)
    A = grid_matrix(shape, drop, diagonal)
    n = A.shape[0]
    extra = 0.0
    if not diagonal and drop == 0.0 and rng.integers(0, 3) == 0:
        # bipartite but for a few odd cycles: an extra coupling (a grid diagonal) at a small share of the vertices
        extra = float(rng.choice([0.002, 0.01, 0.05]))
        stride = shape[-1] + 1
        v = rng.choice(n - stride - 1, max(1, int(extra * n)), replace=False)
        v = v[(v % shape[-1]) < shape[-1] - 1]
        if len(shape) == 3 and rng.integers(0, 4):      # (one case in four keeps the links that wrap into the next plane: odd cycles
            v = v[((v // shape[-1]) % shape[-2]) < shape[-2] - 1]     # without a triangle -- the repair gives up, greedy colouring)
        E = sp.coo_matrix((-rng.uniform(0.1, 0.5, v.size), (v, v + stride)), shape=(n, n)).tocsr()
        E = E + E.T
        A = (A + E + sp.diags(np.asarray(abs(E).sum(axis=1)).ravel())).tocsr()
        A.sort_indices()
    scramble = bool(rng.integers(0, 3) == 0)
    if scramble:
        p = rng.permutation(n)
        A = A[p][:, p].tocsr()
        A.sort_indices()
    reorder = str(rng.choice(["auto", "None", "rcm"])) if scramble else "None"
    reorder = None if reorder == "None" else reorder
    tag = f"case {case}: shape={shape} n={n} drop={drop} diagonal={diagonal} extra={extra} scramble={scramble} reorder={reorder}"
    S = D.CsrSystem.from_any(A, reorder=reorder)
    S.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    refresh = bool(rng.integers(0, 2))
    if refresh:
        # the time-stepping sequence: other values on the same pattern (the factor is parked, the setup only computes values -- the
        # first time it also builds its entry maps), then the original values again; everything below must hold as for a new handle
        d = rng.uniform(0.5, 2.0, n)
        A2 = (sp.diags(d) @ A @ sp.diags(d)).tocsr()
        A2.sort_indices()
        assert np.array_equal(A2.indices, A.indices)
        for vals in (A2.data, A.data, A2.data, A.data)[:int(rng.choice([2, 4]))]:
            S.update_values(vals if rng.integers(0, 2) else torch.from_numpy(vals).cuda())
            S.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    tag += f" refresh={refresh}"
    nc, q = S.precond_ordering()
    Bc = A[q][:, q].tocsr()
    Bc.sort_indices()
    Lref = CO.ic0(Bc)
    rp, ci, v = S.factor()
    ok = np.array_equal(rp, Lref.indptr) and np.array_equal(ci, Lref.indices) and np.array_equal(v, Lref.data)
    what = "factor"
    if ok:
        b = rng.uniform(-1, 1, n)
        zc = CO.sptrsv_upper(CO.transpose_csr(Lref), CO.sptrsv_lower(Lref, b[q]))
        zref = np.empty(n)
        zref[q] = zc
        bd = torch.from_numpy(b).cuda()
        ok = (np.array_equal(S.precond_apply(bd).cpu().numpy(), zref) and np.array_equal(S.precond_apply(bd).cpu().numpy(), zref))
        what = "apply"
    if ok:
        qinv = np.empty(n, dtype=np.int32)
        qinv[q] = np.arange(n, dtype=np.int32)
        if S.reordered:
            ph = S.permutation()
            B = A[ph][:, ph].tocsr()
            B.sort_indices()
            bb, pperm = b[ph], qinv[ph]
        else:
            B, bb, pperm = A, b, qinv
        _, it, hist, _ = CO.pcg(B, bb, "llt_solve", L=Lref, precond_perm=pperm, max_iter=60)
        flags = 0 if rng.integers(0, 2) else D._lib.NO_GRAPH
        res = S.solve(bd, max_iter=60, flags=flags)
        m = min(len(hist), len(res.res_history))
        ok = res.iterations == it and np.allclose(res.res_history[:m], hist[:m], rtol=1e-10, atol=0)
        what = f"solve (iterations {res.iterations} vs {it})"
        # ... and with the oracle's dot products in the device's reduction trees (the colour sweeps' <r,z> included): bit for bit
        geo = S.reduction_geometry()
        ci = S.chip_info()
        if ok and ci["chip_by_default"] and flags == 0:
            # (round 6) the plain call was the one-launch kernel with the triangular solves inside: the whole-chip tree
            tree = {**geo, "form": "chip", "rows_per_workgroup": ci["rows_per_workgroup"]}
            _, it_t, hist_t, _ = CO.pcg(B, bb, "llt_solve", L=Lref, precond_perm=pperm, max_iter=60, device_tree=tree)
            ok = res.iterations == it_t and np.array_equal(res.res_history, hist_t)
            what = f"solve against the whole-chip-tree oracle (iterations {res.iterations} vs {it_t})"
            bitwise += 1
        elif ok and geo["rz_kind"] in (1, 2, 4) and geo["spmv_kernel"] != "vector" and (n > 6144 or (flags & D._lib.NO_SMALL)):
            if geo["rz_kind"] == 4:
                hidx = q
                if S.reordered:
                    iph = np.empty(n, dtype=np.int64)
                    iph[S.permutation()] = np.arange(n)
                    hidx = iph[q]
                geo["sweep_rows"] = CO.sweep_rows(Lref, hidx)
            _, it_t, hist_t, _ = CO.pcg(B, bb, "llt_solve", L=Lref, precond_perm=pperm, max_iter=60, device_tree=geo)
            ok = res.iterations == it_t and np.array_equal(res.res_history, hist_t)
            what = f"solve against the device-tree oracle (rz_kind {geo['rz_kind']}, iterations {res.iterations} vs {it_t})"
            bitwise += 1
        # the apply after a solve (the loop's hand-over of the first level must not leave anything behind)
        ok = ok and np.array_equal(S.precond_apply(bd).cpu().numpy(), zref)
    info = S.info()
    if not ok:
        bad += 1
        print("MULTICOLOUR MISMATCH in", what, tag, flush=True)
    elif len(sys.argv) > 3:
        print("ok", tag, "colours", nc, "levels", info["levels_lower"], info["levels_upper"], flush=True)
    S.close()
print(f"fuzz_multicolour: {cases} cases ({bitwise} of them also bit for bit against the device-tree oracle), {bad} mismatches")
sys.exit(1 if bad else 0)
