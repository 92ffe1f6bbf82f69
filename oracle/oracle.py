"""CPU oracle for the preconditioned-CG hot path -- TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy restatement of the reference algorithm
(`uibk/deep_preconditioning/cg.py`, `utils.py`, and the preconditioner constructors of
`test.py`).  It is the checker the HIP path is compared against.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the product
package `deeppreconditioning_amd` never does.

Parity status: PINNED.  Every function here that restates an importable reference function
(`cg.py`, `utils.py`) is checked in `tests/test_oracle_golden.py` against fixtures in
`tests/golden/` that were produced by importing the reference itself in the build container
(`tests/golden/make_golden.py`).  The preconditioner *constructors* that need ilupp / pyamg /
spconv (absent, `test.py:81-105`) are unpinned against those binaries: `ic0` (textbook IC(0)) and `icholt` (ILU++'s
dual-threshold rule, the harness's default `ilupp.icholt(add_fill_in=1, threshold=0.1)`) restate the PUBLISHED algorithms
and are pinned by their properties (tests/test_oracle_golden.py); see DESIGN.md.  The mesh generators
(`quadtree_fv_laplacian`, `delaunay_laplacian`) are inputs, not algorithm: they stand in for the OpenFOAM matrices of
BASELINE config 3.  The C twin (pcg_oracle.c) can also add its dot products in the DEVICE's reduction trees
(`c_oracle.pcg(..., device_tree=...)`): that changes the order of additions only, and makes the comparison bit for bit.

Every function cites the reference file:line it follows (paths relative to the reference root).
"""

from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# --------------------------------------------------------------------------------------------
# Synthetic inputs.  The reference cannot generate these sizes itself
# (its matrices come from OpenFOAM, generate_data.py:55-81); the generators are closed-form.
# --------------------------------------------------------------------------------------------


def poisson2d(n: int, dtype=np.float64) -> sp.csr_matrix:
    """5-point Laplacian kron(I,T)+kron(T,I), T=tridiag(-1,2,-1): diag 4, off-diag -1.

    Row i = iy*n + ix; columns sorted ascending (i-n, i-1, i, i+1, i+n).  int32 indices.
    """
    idx = np.arange(n * n, dtype=np.int64)
    ix = idx % n
    iy = idx // n
    cols = np.stack([idx - n, idx - 1, idx, idx + 1, idx + n], axis=1)
    mask = np.stack([iy > 0, ix > 0, np.ones_like(ix, bool), ix < n - 1, iy < n - 1], axis=1)
    vals = np.broadcast_to(np.array([-1.0, -1.0, 4.0, -1.0, -1.0], dtype=dtype), cols.shape)
    rowptr = np.zeros(n * n + 1, dtype=np.int32)
    np.cumsum(mask.sum(axis=1), out=rowptr[1:])
    return sp.csr_matrix(
        (vals[mask].astype(dtype), cols[mask].astype(np.int32), rowptr), shape=(n * n, n * n)
    )


def poisson3d(n: int, dtype=np.float64) -> sp.csr_matrix:
    """7-point Laplacian on an n^3 grid: diag 6, off-diag -1.  Row i = (iz*n + iy)*n + ix."""
    N = n * n * n
    idx = np.arange(N, dtype=np.int64)
    ix = idx % n
    iy = (idx // n) % n
    iz = idx // (n * n)
    cols = np.stack([idx - n * n, idx - n, idx - 1, idx, idx + 1, idx + n, idx + n * n], axis=1)
    mask = np.stack(
        [iz > 0, iy > 0, ix > 0, np.ones_like(ix, bool), ix < n - 1, iy < n - 1, iz < n - 1], axis=1
    )
    vals = np.broadcast_to(np.array([-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0], dtype=dtype), cols.shape)
    rowptr = np.zeros(N + 1, dtype=np.int32)
    np.cumsum(mask.sum(axis=1), out=rowptr[1:])
    return sp.csr_matrix((vals[mask].astype(dtype), cols[mask].astype(np.int32), rowptr), shape=(N, N))


def unstructured_like(A: sp.csr_matrix, seed: int = 0) -> sp.csr_matrix:
    """Stand-in for an OpenFOAM pressure matrix: D (P A P^T) D with a seeded random symmetric
    permutation P and SPD diagonal scaling D = diag(U(0.5, 2)).

    Column indices are sorted within each row (canonical CSR), int32.
    """
    rng = np.random.default_rng(seed)
    n = A.shape[0]
    perm = rng.permutation(n)
    d = rng.uniform(0.5, 2.0, n)
    B = A.tocsr()[perm][:, perm]
    B = sp.diags(d) @ B @ sp.diags(d)
    B = B.tocsr()
    B.sort_indices()
    B.indices = B.indices.astype(np.int32)
    B.indptr = B.indptr.astype(np.int32)
    return B


# --------------------------------------------------------------------------------------------
# Unstructured stand-ins for the OpenFOAM pressure matrices (BASELINE config 3): a quadtree-refined finite-volume
# Laplacian and a Delaunay graph Laplacian.  The reference's matrices come out of interFoam on a snappyHexMesh-refined
# hex grid (generate_data.py:55-81, foam/sim/system/snappyHexMeshDict); OpenFOAM is absent, so these generators make
# matrices of that kind.  The product package has its own copy (deeppreconditioning_amd/meshes.py -- it may not import
# this module); tests/test_meshes.py holds the two to the same bits.
# --------------------------------------------------------------------------------------------


def _mesh_canonical(B: sp.csr_matrix) -> sp.csr_matrix:
    B = B.tocsr()
    B.sum_duplicates()
    B.sort_indices()
    B.indices = B.indices.astype(np.int32)
    B.indptr = B.indptr.astype(np.int32)
    return B


def _blob_distance(m: int, rng, n_blobs: int, r_lo: float, r_hi: float, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """min_k (|(x, y) - c_k| - r_k) for n_blobs seeded circles in [0, m]^2, evaluated blob by blob on the bounding
    box of each circle (+ 2 cells) only: O(sum r^2), not O(n_blobs * m^2)."""
    phi = np.full(x.shape, np.inf)
    cx = rng.uniform(0.0, m, n_blobs)
    cy = rng.uniform(0.0, m, n_blobs)
    rr = rng.uniform(r_lo, r_hi, n_blobs)
    # x, y are the centres of an s x s lattice over [0, m]^2 (s = x.shape[0]); cell pitch m / s
    s = x.shape[0]
    pitch = m / s
    for k in range(n_blobs):
        pad = rr[k] + 2.0
        i0 = max(int((cy[k] - pad) / pitch), 0)
        i1 = min(int((cy[k] + pad) / pitch) + 1, s)
        j0 = max(int((cx[k] - pad) / pitch), 0)
        j1 = min(int((cx[k] + pad) / pitch) + 1, s)
        if i0 >= i1 or j0 >= j1:
            continue
        d = np.hypot(x[i0:i1, j0:j1] - cx[k], y[i0:i1, j0:j1] - cy[k]) - rr[k]
        np.minimum(phi[i0:i1, j0:j1], d, out=phi[i0:i1, j0:j1])
    return phi


def quadtree_fv_laplacian(m: int, seed: int = 0, *, n_blobs: int | None = None, radius=(3.0, 10.0),
                          numbering: str = "foam", return_info: bool = False):
    """Finite-volume Laplacian on an m x m hex grid refined once (2:1, hanging nodes) around random blobs.

    Mesh (what `snappyHexMeshDict` does to `blockMeshDict`'s grid): `n_blobs` seeded circles (default m*m/700) of radius
    U(radius) in cell units; every coarse cell within one cell of a blob's surface is split into 2 x 2 children (level
    (1 1), one buffer layer); cells whose centre lies inside a blob are removed; of what remains the largest
    face-connected region is kept (`locationInMesh`).  Matrix (`pEqn.H:43-46`, sign as `generate_data.py:71`): for
    every face f between cells i, j the coefficient  c_f = k_f |S_f| / (n . d_ij)  (two-point flux, no non-orthogonal
    correction -- `fvSolution` nNonOrthogonalCorrectors 0): |S_f| = 1 (coarse-coarse), 1/2 otherwise; n . d = 1, 3/4
    (coarse-fine), 1/2 (fine-fine); k_f = harmonic mean of seeded cell values U(0.5, 2) (the rAUf field);
    A_ij = -c_f, A_ii = sum_f c_f + Dirichlet closure on the OUTER boundary (k_i |S_f| / (h_i / 2)); blob walls are
    zero-gradient (nothing added).  numbering: "foam" = coarse cells row-major, first child keeps the parent's label, the
    other three are appended in parent order, removed cells compacted away (hexRef8 / removeCells); "random" = a seeded
    random permutation of that.  ~m*m*1.1 rows at the defaults."""
    rng = np.random.default_rng(seed)
    if n_blobs is None:
        n_blobs = max(1, (m * m) // 700)
    cc = np.arange(m) + 0.5
    X, Y = np.meshgrid(cc, cc)                      # [iy, ix]
    phi_c = _blob_distance(m, rng, n_blobs, radius[0], radius[1], X, Y)
    fc = (np.arange(2 * m) + 0.5) * 0.5
    XF, YF = np.meshgrid(fc, fc)
    rng_f = np.random.default_rng(seed)             # same blobs on the fine lattice
    phi_f = _blob_distance(m, rng_f, n_blobs, radius[0], radius[1], XF, YF)
    refined = np.abs(phi_c) < 1.5                   # cut cells + one buffer layer
    # labels, hexRef8 style: coarse cell (iy, ix) has label iy*m + ix; child 0 (lower left) keeps it, children 1..3
    # get m*m + 3*rank(parent) + (0, 1, 2)
    parent_rank = np.cumsum(refined.ravel()) - 1
    base = (np.arange(m * m)).reshape(m, m)
    owner = np.repeat(np.repeat(base, 2, axis=0), 2, axis=1).astype(np.int64)     # fine lattice -> label
    ref_f = np.repeat(np.repeat(refined, 2, axis=0), 2, axis=1)
    child = (np.arange(2 * m)[:, None] % 2) * 2 + (np.arange(2 * m)[None, :] % 2)   # 0..3 inside the parent
    pr_f = np.repeat(np.repeat(parent_rank.reshape(m, m), 2, axis=0), 2, axis=1)
    extra = ref_f & (child > 0)
    owner[extra] = m * m + 3 * pr_f[extra] + (child[extra] - 1)
    # removal: a fine cell goes by its own centre, a coarse cell by its centre
    phi_owner = np.where(ref_f, phi_f, np.repeat(np.repeat(phi_c, 2, axis=0), 2, axis=1))
    alive = phi_owner >= 0.0
    owner[~alive] = -1
    n_lab = m * m + 3 * int(refined.sum())
    is_fine = np.zeros(n_lab, dtype=bool)
    is_fine[owner[ref_f & alive]] = True
    kappa = rng.uniform(0.5, 2.0, n_lab)

    def faces(a, b):
        ok = (a >= 0) & (b >= 0) & (a != b)
        a, b = a[ok], b[ok]
        fa, fb = is_fine[a], is_fine[b]
        dist = np.where(fa & fb, 0.5, np.where(fa | fb, 0.75, 1.0))
        kf = 2.0 * kappa[a] * kappa[b] / (kappa[a] + kappa[b])
        return a, b, kf * 0.5 / dist                # every lattice edge carries a half face (|S| = 1/2)

    ah, bh, ch = faces(owner[:, :-1].ravel(), owner[:, 1:].ravel())
    av, bv, cv = faces(owner[:-1, :].ravel(), owner[1:, :].ravel())
    a = np.concatenate([ah, av])
    b = np.concatenate([bh, bv])
    c = np.concatenate([ch, cv])
    # Dirichlet closure on the outer boundary: half face over half the cell's width
    edge = np.concatenate([owner[0, :], owner[-1, :], owner[:, 0], owner[:, -1]])
    edge = edge[edge >= 0]
    dir_c = kappa[edge] * 0.5 / np.where(is_fine[edge], 0.25, 0.5)
    diag = np.zeros(n_lab)
    np.add.at(diag, a, c)
    np.add.at(diag, b, c)
    np.add.at(diag, edge, dir_c)
    A = sp.coo_matrix((np.concatenate([-c, -c]), (np.concatenate([a, b]), np.concatenate([b, a]))),
                      shape=(n_lab, n_lab)).tocsr()
    A.sum_duplicates()
    # keep the largest face-connected region among the labels that exist
    exists = np.zeros(n_lab, dtype=bool)
    exists[owner[owner >= 0]] = True
    ncomp, comp = sp.csgraph.connected_components(A, directed=False)
    sizes = np.bincount(comp[exists], minlength=ncomp)
    keep = exists & (comp == int(np.argmax(sizes)))
    labels = np.flatnonzero(keep)                   # ascending: compaction keeps the order
    A = A[labels][:, labels] + sp.diags(diag[labels])
    n = labels.size
    if numbering == "random":
        perm = np.random.default_rng(seed + 1).permutation(n)
        A = A.tocsr()[perm][:, perm]
    elif numbering != "foam":
        raise ValueError(numbering)
    A = _mesh_canonical(A)
    if return_info:
        return A, {"coarse": int((~is_fine[labels]).sum()), "fine": int(is_fine[labels].sum()),
                   "blobs": int(n_blobs), "removed": int(n_lab - n)}
    return A


def delaunay_laplacian(n_points: int, seed: int = 0):
    """Weighted graph Laplacian of the 2-D Delaunay triangulation of `n_points` seeded uniform points in the unit square
    (`scipy.spatial.Delaunay`), rows in the order the points were drawn (no spatial order at all).  Edge weight
    w_ij = harmonic mean of seeded vertex values U(0.5, 2); A_ij = -w_ij; A_ii = sum_j w_ij, and every vertex of the convex
    hull gets its diagonal doubled (a Dirichlet ghost neighbour per hull edge): symmetric, irreducibly diagonally dominant,
    positive definite -- the sign convention of `generate_data.py:71-79`."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = rng.random((n_points, 2))
    kappa = rng.uniform(0.5, 2.0, n_points)
    tri = Delaunay(pts)
    s = tri.simplices.astype(np.int64)
    e = np.concatenate([s[:, [0, 1]], s[:, [1, 2]], s[:, [0, 2]]])
    e.sort(axis=1)
    key = np.unique(e[:, 0] * n_points + e[:, 1])
    a, b = key // n_points, key % n_points
    w = 2.0 * kappa[a] * kappa[b] / (kappa[a] + kappa[b])
    diag = np.zeros(n_points)
    np.add.at(diag, a, w)
    np.add.at(diag, b, w)
    hull = np.unique(tri.convex_hull)
    diag[hull] *= 2.0
    A = sp.coo_matrix((np.concatenate([-w, -w, diag]),
                       (np.concatenate([a, b, np.arange(n_points)]), np.concatenate([b, a, np.arange(n_points)]))),
                      shape=(n_points, n_points))
    return _mesh_canonical(A)


def rhs(n: int, seed: int = 0) -> np.ndarray:
    """b ~ U(-1, 1), the distribution of generate_data.py:106, seeded as SURVEY.md 8-c3."""
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n)


# --------------------------------------------------------------------------------------------
# Operators
# --------------------------------------------------------------------------------------------


def spmv(A: sp.csr_matrix, x: np.ndarray) -> np.ndarray:
    """y = A x, CSR row sums in column order (what `A @ pk` means at cg.py:75 for a CSR A).

    scipy's csr_matvec accumulates each row sequentially `sum += Ax[jj] * Xx[Aj[jj]]` with no
    FMA contraction on the x86-64 baseline build; the HIP kernel reproduces exactly that order
    and rounding, so SpMV parity is bit-exact.
    """
    return A @ x


def jacobi_dinv(A: sp.csr_matrix) -> np.ndarray:
    """Diagonal of the Jacobi preconditioner, `1 / matrix.diagonal()` (test.py:74-79)."""
    return 1.0 / A.diagonal()


def ic0(A: sp.csr_matrix) -> sp.csr_matrix:
    """Zero-fill incomplete Cholesky factor L (lower triangular CSR, pattern = tril(A)).

    Stands in for `ilupp.ichol0` (test.py:83; ilupp is absent, so this is the textbook IC(0),
    Saad 2003 Alg. 10.x, row-oriented):  for each row i, for each stored j<i ascending:
        L_ij = (a_ij - sum_{m<j, m in pat(i) & pat(j)} L_im L_jm) / L_jj
        L_ii = sqrt(a_ii - sum_{m<i} L_im^2)
    Sums run over ascending m, one product at a time, no FMA -- the native setup routine
    follows the same order so the factors agree bit for bit.
    """
    T = sp.tril(A, format="csr")
    T.sort_indices()
    n = T.shape[0]
    rp, ci = T.indptr, T.indices
    lv = T.data.astype(np.float64).copy()
    for i in range(n):
        s_i, e_i = rp[i], rp[i + 1]
        for k in range(s_i, e_i):
            j = ci[k]
            s_j, e_j = rp[j], rp[j + 1]
            acc = lv[k]
            # two-pointer intersection of row i and row j restricted to columns < j
            a, b = s_i, s_j
            while a < k and b < e_j - 1:
                ca, cb = ci[a], ci[b]
                if ca == cb:
                    acc = acc - lv[a] * lv[b]
                    a += 1
                    b += 1
                elif ca < cb:
                    a += 1
                else:
                    b += 1
            if j < i:
                lv[k] = acc / lv[e_j - 1]  # diagonal is the last entry of a tril row
            else:
                lv[k] = np.sqrt(acc)
    return sp.csr_matrix((lv, ci.copy(), rp.copy()), shape=T.shape)


def ict(A: sp.csr_matrix, fill_in: int = 1, threshold: float = 0.1, row_cap: int = 192) -> sp.csr_matrix:
    """Thresholded incomplete Cholesky with level-1 fill: the contract of `CsrSystem.set_preconditioner(ICT(...))`.

    Stands in for `ilupp.icholt(A, add_fill_in=1, threshold=0.1)`, the reference harness's DEFAULT incomplete-Cholesky
    technique (test.py:81-88).  ilupp is absent (not vendored, no network) and publishes no test vectors: PARITY
    UNPINNED.  The algorithm restated here is the textbook one -- level-of-fill symbolic phase (Saad 2003, IC(p)) with the
    drop rule MATLAB documents for `ichol(..., type='ict')`:
      pattern  S_i = {j <= i : a_ij != 0}  plus, for fill_in >= 1, {j < i : there is k < j with a_ik != 0 and a_jk != 0}
               (fill created by eliminating with ORIGINAL entries only; fill_in > 1 is treated as 1; a row whose
               pattern would then exceed `row_cap` = 192 entries keeps tril(A)_i -- the bound of the device routine's
               private per-row set, part of the contract);
      numeric  row by row, stored columns ascending:  acc = a_ij (0 for a fill position) - sum_{m<j, m in S_i & S_j} L_im L_jm
               (ascending m, one product and one subtraction at a time);  j < i:  v = acc / L_jj, and v is DROPPED (stored as
               0, so later sums see 0) when |v| * L_jj < threshold * ||A(j:n, j)||_1;  j = i:  L_ii = sqrt(acc) > 0;
      result   L without the dropped entries, columns ascending, diagonal last.
    threshold = 0 and fill_in = 0 give IC(0) exactly (same operation order as `ic0`)."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    arp, aci, av = A.indptr, A.indices, A.data.astype(np.float64)
    colnorm = np.zeros(n)
    for j in range(n):
        for k in range(arp[j], arp[j + 1]):
            if aci[k] >= j:
                colnorm[j] += abs(av[k])
    rp = [0]
    ci, lv = [], []
    for i in range(n):
        row = {int(aci[k]): float(av[k]) for k in range(arp[i], arp[i + 1]) if aci[k] <= i}
        if len(row) > row_cap:
            raise ValueError("ICT: a row of tril(A) has more than row_cap entries")
        if fill_in >= 1:
            filled = dict(row)
            for k in [c for c in row if c < i]:
                for q in range(arp[k], arp[k + 1]):
                    j = int(aci[q])
                    if k < j < i and j not in filled:
                        filled[j] = 0.0
            if len(filled) <= row_cap:
                row = filled
        for c in sorted(row):
            ci.append(c)
            lv.append(row[c])
        rp.append(len(ci))
    rp, ci, lv = np.array(rp), np.array(ci), np.array(lv, dtype=np.float64)
    for i in range(n):
        s_i, e_i = rp[i], rp[i + 1]
        for k in range(s_i, e_i):
            j = ci[k]
            s_j, e_j = rp[j], rp[j + 1]
            acc = lv[k]
            a, b = s_i, s_j
            while a < k and b < e_j - 1:
                ca, cb = ci[a], ci[b]
                if ca == cb:
                    acc = acc - lv[a] * lv[b]
                    a += 1
                    b += 1
                elif ca < cb:
                    a += 1
                else:
                    b += 1
            if j < i:
                d = lv[e_j - 1]
                v = acc / d
                lv[k] = 0.0 if abs(v) * d < threshold * colnorm[j] else v
            else:
                if not acc > 0.0:
                    raise ValueError(f"ICT: non-positive pivot at row {i}")
                lv[k] = np.sqrt(acc)
    keep = (lv != 0.0) | (ci == np.repeat(np.arange(n), np.diff(rp)))
    out_rp = np.concatenate(([0], np.cumsum(np.add.reduceat(keep.astype(np.int64), rp[:-1]))))
    return sp.csr_matrix((lv[keep], ci[keep], out_rp), shape=A.shape)


def icholt(A: sp.csr_matrix, add_fill_in: int = 0, threshold: float = 0.0, cand_cap: int = 256, row_cap: int = 64) -> sp.csr_matrix:
    """Thresholded incomplete Cholesky as ILU++ defines it -- the contract of `ilupp.icholt(A, add_fill_in, threshold)`, the
    reference harness's DEFAULT incomplete-Cholesky technique (test.py:81-88: `icholt(matrix, add_fill_in=1, threshold=0.1)`).

    ilupp 1.0.2 (uv.lock:952) wraps ILU++ (J. Mayer, "ILU++: a new software package for solving sparse linear systems with
    iterative methods", PAMM 7 (2007); its thresholded factorisations follow Y. Saad's dual-threshold ILUT(p, tau), Numer.
    Linear Algebra Appl. 1 (1994), restricted to the lower triangle of a symmetric matrix).  The package is absent from the
    image and publishes no test vectors, so the PUBLISHED algorithm is restated (PARITY UNPINNED against the binary; the two
    choices the papers leave open -- which norm, how ties break -- are fixed below and are part of this contract):

      for k = 0 .. n-1 (left-looking, one column of L at a time):
        w      = A[k:, k]                                            the lower part of column k (fill positions start at 0)
        for every j < k with L[k, j] kept, ASCENDING j:              w[i] -= L[k, j] * L[i, j]  for the kept i >= k of column j
                                                                     (one product, one subtraction: two roundings)
        d      = sqrt(w[k])                                          (w[k] <= 0: breakdown, ValueError)
        norm   = sqrt(sum_i w[i]^2) over the off-diagonal candidates i > k, summed in ascending i
        drop   every candidate with |w[i]| < threshold * norm        ("entries with a relative magnitude less than this")
        keep   of the rest the  p_k = nnz(A[k+1:, k]) + add_fill_in  largest in magnitude (ties: the smaller row index)
        L[i, k] = w[i] / d  for the kept i,   L[k, k] = d

    threshold = 0 keeps the p_k largest (IC with `add_fill_in` extra entries per column); add_fill_in = 0, threshold = 0 on a
    matrix whose IC(0) creates no fill-free cancellation is NOT IC(0): ICT picks the largest entries, which may be fill.
    `cand_cap` / `row_cap` are the bounds of the device routine (candidates per column, kept entries per row of L): exceeding
    them raises here as it fails there.  Returns L as lower-triangular CSR, columns ascending, diagonal last."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    arp, aci, av = A.indptr, A.indices, A.data.astype(np.float64)
    rows_j = [[] for _ in range(n)]      # row i of L: the kept (j, L_ij), ascending j by construction
    cols = [None] * n                    # column j of L: kept (i, L_ij), ascending i
    for k in range(n):
        cand = {}
        diag = 0.0
        p_k = 0
        for q in range(arp[k], arp[k + 1]):          # row k of the symmetric A = column k: entries with index >= k
            i = int(aci[q])
            if i == k:
                diag = float(av[q])
            elif i > k:
                cand[i] = float(av[q])
                p_k += 1
        p_k += int(add_fill_in)
        if p_k > row_cap:
            raise ValueError("icholt: nnz + add_fill_in of a column exceeds row_cap")
        for j, lkj in rows_j[k]:
            diag = diag - lkj * lkj
            for i, lij in cols[j]:
                if i > k:
                    cand[i] = cand.get(i, 0.0) - lkj * lij
        if len(cand) > cand_cap:
            raise ValueError("icholt: more candidates in a column than cand_cap")
        if not diag > 0.0:
            raise ValueError(f"icholt: non-positive pivot at column {k}")
        d = float(np.sqrt(diag))
        items = sorted(cand.items())
        ss = 0.0
        for _, v in items:
            ss = ss + v * v
        norm = float(np.sqrt(ss))
        kept = [(i, v) for i, v in items if not abs(v) < threshold * norm]
        if len(kept) > p_k:
            kept = sorted(sorted(kept, key=lambda t: (-abs(t[1]), t[0]))[:p_k])
        col = [(i, v / d) for i, v in kept]
        cols[k] = col
        for i, lik in col:
            if len(rows_j[i]) >= row_cap:
                raise ValueError("icholt: more kept entries in a row than row_cap")
            rows_j[i].append((k, lik))
        rows_j[k].append((k, d))
    rp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum([len(r) for r in rows_j], out=rp[1:])
    ci = np.fromiter((j for r in rows_j for j, _ in r), dtype=np.int32, count=int(rp[-1]))
    lv = np.fromiter((v for r in rows_j for _, v in r), dtype=np.float64, count=int(rp[-1]))
    return sp.csr_matrix((lv, ci, rp), shape=A.shape)


def learned_like_factor(A: sp.csr_matrix, seed: int = 0, scale: float = 0.05, diag_sigma: float = 1.0) -> sp.csr_matrix:
    """A seeded stand-in for the CNN output L = PreconditionerNet(tril(A)) (model.py:42-59).

    Pattern: tril(A)'s pattern dilated by offsets [-2,2]x[-2,2] (the Minkowski sum of the four
    2x2 sparse convolutions, model.py:33-37), clipped to the matrix and to the lower triangle
    (model.py:53-54 zeroes the strict upper part; `to_sparse_csr` drops it, test.py:105).
    Values: strict-lower ~ scale*N(0,1) in fp32, diagonal = softplus(diag_sigma*N(0,1)) > 0 (model.py:56-57),
    upcast to fp64 as the reference does (test.py:105).  No checkpoint or spconv is available,
    so the values are random; only the algebraic contract is honoured (parity unpinned).
    """
    rng = np.random.default_rng(seed)
    T = sp.tril(A, format="coo")
    n = A.shape[0]
    rows, cols = [], []
    for dr in range(-2, 3):
        for dc in range(-2, 3):
            r = T.row.astype(np.int64) + dr
            c = T.col.astype(np.int64) + dc
            ok = (r >= 0) & (r < n) & (c >= 0) & (c < n) & (c <= r)
            rows.append(r[ok])
            cols.append(c[ok])
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    key = np.unique(rows * n + cols)
    rows, cols = key // n, key % n
    vals = (scale * rng.standard_normal(len(key))).astype(np.float32)
    diag = rows == cols
    g = (diag_sigma * rng.standard_normal(int(diag.sum()))).astype(np.float32)
    vals[diag] = np.log1p(np.exp(g))  # softplus
    L = sp.csr_matrix((vals.astype(np.float64), (rows, cols)), shape=(n, n))
    L.sort_indices()
    L.indices = L.indices.astype(np.int32)
    L.indptr = L.indptr.astype(np.int32)
    return L


def learned_like_factor_preconditioning(A: sp.csr_matrix, seed: int = 1, noise: float = 0.005) -> sp.csr_matrix:
    """A learned-like factor that really preconditions -- the stand-in for a TRAINED PreconditionerNet output
    (`learned_like_factor` above is the untrained, random-weight one, whose M A is so ill-conditioned that the
    multiplied PCG is chaotic: the reference marks the technique `# unstable`, test.py:45).

    L = (I + E)/2 + noise * N(0,1) on the strict-lower part of the CNN's output pattern (tril(A) dilated by
    [-2,2]^2, model.py:33-37), rounded to fp32 and upcast (test.py:105); E = -strict_lower(A)/diag(A).
    (I + E)(I + E^T)/4 is the first-order Neumann approximation of the symmetric Gauss-Seidel inverse of A (exact
    orientation for matrices that are invariant under reversing the numbering, like the grid Laplacians).
    M = L L^T multiplied gives ~half of Jacobi's iteration count and a numerically STABLE recurrence, so BASELINE
    config 2 ("CNN-emitted L factor") gets a count-exact 1e-10 fixture from the reference's own loop."""
    n = A.shape[0]
    d = A.diagonal()
    E = (-sp.tril(A, -1).tocsr()).multiply(1.0 / d[:, None]).tocsr()
    R = learned_like_factor(A, seed=seed, scale=noise, diag_sigma=0.0)
    R = (R - sp.diags(R.diagonal())).tocsr()
    L = (0.5 * (sp.identity(n, format="csr") + E) + R).tocsr()
    L.data = L.data.astype(np.float32).astype(np.float64)
    L.sort_indices()
    L.indices = L.indices.astype(np.int32)
    L.indptr = L.indptr.astype(np.int32)
    return L


def sptrsv_lower(L: sp.csr_matrix, r: np.ndarray) -> np.ndarray:
    """Solve L y = r by forward substitution, row sums in column order, then divide by L_ii."""
    rp, ci, lv = L.indptr, L.indices, L.data
    y = np.zeros_like(r, dtype=np.float64)
    for i in range(L.shape[0]):
        acc = r[i]
        e = rp[i + 1] - 1
        for k in range(rp[i], e):
            acc = acc - lv[k] * y[ci[k]]
        y[i] = acc / lv[e]
    return y


def sptrsv_upper_t(L: sp.csr_matrix, y: np.ndarray) -> np.ndarray:
    """Solve L^T z = y by backward substitution on U = L^T stored as CSR (diagonal first in a
    row, remaining columns ascending), row sums in column order, then divide by U_ii."""
    U = L.T.tocsr()
    U.sort_indices()
    rp, ci, uv = U.indptr, U.indices, U.data
    z = np.zeros_like(y, dtype=np.float64)
    for i in range(U.shape[0] - 1, -1, -1):
        acc = y[i]
        s = rp[i]
        for k in range(s + 1, rp[i + 1]):
            acc = acc - uv[k] * z[ci[k]]
        z[i] = acc / uv[s]
    return z


class Precond:
    """The ways `zk = M @ rk` (cg.py:61,81) is realised.  kinds:

    none          M = I                                 (test.py:70-72)
    jacobi        M = diag(1/a_ii)                      (test.py:74-79)
    csr           M given as CSR, z = M r               (test.py:88,105: M = L L^T materialised)
    llt_multiply  z = L (L^T r) without forming L L^T   (same operator as test.py:102-105)
    llt_solve     z = L^-T (L^-1 r), true IC apply      (north_star; not in the reference)
    """

    def __init__(self, kind: str, *, dinv=None, M=None, L=None):
        self.kind = kind
        self.dinv = dinv
        self.M = M
        self.L = L
        self.Lt = L.T.tocsr() if L is not None else None
        if self.Lt is not None:
            self.Lt.sort_indices()

    def __matmul__(self, r: np.ndarray) -> np.ndarray:
        if self.kind == "none":
            return r.copy()
        if self.kind == "jacobi":
            return self.dinv * r
        if self.kind == "csr":
            return self.M @ r
        if self.kind == "llt_multiply":
            return self.L @ (self.Lt @ r)
        if self.kind == "llt_solve":
            y = spla.spsolve_triangular(self.L, r, lower=True)
            return spla.spsolve_triangular(self.Lt, y, lower=False)
        raise ValueError(self.kind)


class PermutedPrecond:
    """z' = P M P^T r' for a preconditioner M kept in the CALLER's numbering while the system is P A P^T (row i = the
    caller's row perm[i]): what a reordered libdpcg handle does with a factor it solves with (dpcg_reorder)."""

    def __init__(self, M, perm):
        self.M, self.perm = M, np.asarray(perm)

    def __matmul__(self, r: np.ndarray) -> np.ndarray:
        rc = np.empty_like(r)
        rc[self.perm] = r
        return (self.M @ rc)[self.perm]


# --------------------------------------------------------------------------------------------
# Solvers
# --------------------------------------------------------------------------------------------


def stopping_criterion(_, rk: np.ndarray, b: np.ndarray) -> float:
    """<rk,rk>/<b,b>: the SQUARED relative residual (cg.py:15-17)."""
    return float(np.dot(rk, rk) / np.dot(b, b))


def preconditioned_conjugate_gradient(A, b, M, x0=None, rtol=1e-8, max_iter=1024, init_check="z"):
    """Restatement of cg.py:50-90 without the wasted `A @ zeros` of cg.py:85-87.

    Returns (seconds, iterations, residual_history, x).  `residual_history[k]` is the value the
    reference appends to `errors` (cg.py:67,88): entry 0 is <z0,z0>/<b,b> (the cg.py:66 quirk:
    the initial check uses zk), entries k>=1 are <rk,rk>/<b,b>.  `init_check="r"` gives the
    scipy-style first test on r instead (utils.py:66-72 path).
    """
    x = np.zeros_like(b, dtype=np.float64) if x0 is None else np.array(x0, dtype=np.float64)  # cg.py:58
    r = b - A @ x  # cg.py:60
    z = M @ r  # cg.py:61
    p = z.copy()  # cg.py:62
    bb = np.dot(b, b)
    res = np.dot(z, z) / bb if init_check == "z" else np.dot(r, r) / bb  # cg.py:66
    hist = [float(res)]
    t0 = time.perf_counter()  # cg.py:69
    for _ in range(max_iter):  # cg.py:70
        if res < rtol:  # cg.py:71
            break
        Ap = A @ p  # cg.py:75
        rz = np.dot(r, z)  # cg.py:76
        a = rz / np.dot(Ap, p)  # cg.py:78
        x = x + a * p  # cg.py:79
        r = r - a * Ap  # cg.py:80
        z = M @ r  # cg.py:81
        beta = np.dot(r, z) / rz  # cg.py:82
        p = z + beta * p  # cg.py:83
        res = np.dot(r, r) / bb  # cg.py:86
        hist.append(float(res))
    t1 = time.perf_counter()  # cg.py:88
    return t1 - t0, len(hist) - 1, np.array(hist), x  # cg.py:90 (+ history and x for the checker)


class MixedOperator:
    """Config 5's `A @ pk` (cg.py:75): matrix values and the vector STORED in fp32, products and row sums in fp64 --
    y = fp64(fp32(A)) @ fp64(fp32(v)).  Handed to `preconditioned_conjugate_gradient` as A it gives the mixed-precision
    PCG (x0 = 0, so the only `A @` that matters is the loop's; everything else stays fp64).  The same duck-typed operator,
    on torch tensors, is what tests/golden/make_golden.py --add-round3 hands the REFERENCE's loop to pin this."""

    def __init__(self, A: sp.csr_matrix):
        A = A.tocsr()
        self.A32 = sp.csr_matrix((A.data.astype(np.float32).astype(np.float64), A.indices, A.indptr), shape=A.shape)
        self.shape = A.shape

    def __matmul__(self, v: np.ndarray) -> np.ndarray:
        return self.A32 @ v.astype(np.float32).astype(np.float64)


class MixedOperatorX0(MixedOperator):
    """`MixedOperator` for a solve with x0 != 0: the FIRST product (the initial residual b - A x0, cg.py:60) is the plain
    fp64 one -- only the loop's `A @ pk` (cg.py:75) is mixed, as in orc_pcg_mixed and DPCG_SPMV_F32."""

    def __init__(self, A: sp.csr_matrix):
        super().__init__(A)
        self.A = A.tocsr()
        self.first = True

    def __matmul__(self, v: np.ndarray) -> np.ndarray:
        if self.first:
            self.first = False
            return self.A @ v
        return super().__matmul__(v)


def ground_truth_solve(A, b, atol=1e-6, maxiter=None):
    """The ground-truth solve of the data generator, generate_data.py:107: `scipy.sparse.linalg.cg(matrix, rhs, rtol=0,
    atol=1e-6)`.  The arithmetic lives in scipy (pinned 1.15.1 in the reference's uv.lock, 1.15.3 in this image; not
    vendored), whose published algorithm is restated here: x0 = 0, stop at the top of an iteration when
    ||r||_2 < max(atol, rtol ||b||), otherwise rho = <r,z> (z = r: no preconditioner), p = z + (rho / rho_prev) p,
    q = A p, alpha = rho / <p,q>, x += alpha p, r -= alpha q; at most 10 n iterations.  Returns (x, iterations, info)
    with iterations = completed updates (= callback calls) and info = 0 on convergence, maxiter otherwise.
    Pinned by tests/golden `ground_truth/*` (outputs of the verbatim scipy call)."""
    n = b.shape[0]
    maxiter = 10 * n if maxiter is None else maxiter
    x = np.zeros(n, dtype=np.float64)
    r = np.array(b, dtype=np.float64)
    p = None
    rho_prev = 0.0
    for it in range(maxiter):
        if np.linalg.norm(r) < atol:
            return x, it, 0
        rho = np.dot(r, r)
        p = r.copy() if it == 0 else r + (rho / rho_prev) * p
        q = A @ p
        alpha = rho / np.dot(p, q)
        x = x + alpha * p
        r = r - alpha * q
        rho_prev = rho
    return x, maxiter, maxiter


def conjugate_gradient(A, b, x0=None, x_true=None, rtol=1e-8, max_iter=1024):
    """Restatement of cg.py:20-47.  Returns (errors, x_hat), errors[k] = (A-norm error or 0, res)."""
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=b.dtype)  # cg.py:22
    r = b - A @ x  # cg.py:23
    p = r.copy()  # cg.py:24
    bb = np.dot(b, b)

    def a_norm_err(xh):
        if x_true is None:
            return 0.0
        e = xh - x_true  # cg.py:27
        return float(np.dot(e, A @ e))  # cg.py:29

    res = np.dot(r, r) / bb  # cg.py:28
    errors = [(a_norm_err(x), float(res))]
    for _ in range(max_iter):  # cg.py:31
        if res < rtol:  # cg.py:32
            break
        Ap = A @ p  # cg.py:35
        r_norm = np.dot(r, r)  # cg.py:36
        a = r_norm / np.dot(Ap, p)  # cg.py:38
        x = x + a * p  # cg.py:39
        r = r - a * Ap  # cg.py:40
        p = r + (np.dot(r, r) / r_norm) * p  # cg.py:41
        res = np.dot(r, r) / bb  # cg.py:44
        errors.append((a_norm_err(x), float(res)))
    return errors, x


def sparse_matvec_mul(indices: np.ndarray, features: np.ndarray, batch_size: int, vector_batch: np.ndarray,
                      transpose: bool) -> np.ndarray:
    """Restatement of utils.py:15-43 on plain arrays: batched COO SpMV / SpMV^T in fp32.

    indices (nnz,3) int32 = (batch,row,col); features (nnz,1) fp32; vector_batch (B, dof) fp32.
    Products are scattered onto rows in storage order (scatter_reduce "sum", utils.py:36-41).
    """
    b_idx = indices[:, 0].astype(np.int64)
    r_idx = indices[:, 2 if transpose else 1].astype(np.int64)  # utils.py:27
    c_idx = indices[:, 1 if transpose else 2].astype(np.int64)  # utils.py:28
    out = np.zeros_like(vector_batch, dtype=np.float32)
    prod = (features[:, 0].astype(np.float32) * vector_batch[b_idx, c_idx].astype(np.float32)).astype(np.float32)
    for k in range(len(prod)):  # sequential fp32 accumulation in storage order
        out[b_idx[k], r_idx[k]] = np.float32(out[b_idx[k], r_idx[k]] + prod[k])
    return out


def benchmark_cg(matrix, right_hand_side, preconditioner=None):
    """Restatement of utils.py:46-76: scipy cg, maxiter=512, default rtol=1e-5, callback count."""
    iterations = 0

    def _callback(_):
        nonlocal iterations
        iterations += 1

    t0 = time.perf_counter()
    _, info = spla.cg(matrix, right_hand_side, maxiter=512, M=preconditioner, callback=_callback)
    return time.perf_counter() - t0, iterations, info
