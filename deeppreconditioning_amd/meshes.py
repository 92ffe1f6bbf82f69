"""Unstructured stand-ins for the reference's OpenFOAM pressure matrices (BASELINE config 3).

The reference's systems are finite-volume pressure Laplacians on a hex mesh that snappyHexMesh has refined once
around an STL surface and cut open (`foam/sim/system/snappyHexMeshDict`: `refinementSurfaces ... level (1 1)`,
`nCellsBetweenLevels 1`; `blockMeshDict`: one cell thick, front/back `empty`), dumped as triplets and sign-flipped to
positive diagonal / negative off-diagonals (`generate_data.py:55-81`, `pEqn.H:98-108`).  OpenFOAM is not in the image,
so two generators produce matrices of that KIND at any size (host-side numpy/scipy: setup-time plumbing, the solve never
touches the host copy):

* `quadtree_fv_laplacian`  -- cell-centred two-point-flux Laplacian on a 2-D hex grid with one level of 2:1 refinement
  (hanging nodes) around random blobs whose interior is removed, numbered the way hexRef8 numbers (first child keeps the
  parent's label, the others are appended) or at random: 3-9 entries per row, triangles at every hanging node, holes.
* `delaunay_laplacian`     -- weighted graph Laplacian of the Delaunay triangulation of random points (numbered in the
  order the points were drawn, i.e. at random): ~7 entries per row, degrees 3-12+, every edge on a triangle.

Both are symmetric positive definite M-matrices (irreducibly diagonally dominant: Dirichlet closure on the outer
boundary), with sorted int32 CSR indices.  `oracle/oracle.py` holds an independent restatement; the CPU tests compare
the two (`tests/test_meshes.py`).
"""

from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def _canonical(B: sp.csr_matrix) -> sp.csr_matrix:
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
    A = _canonical(A)
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
    return _canonical(A)
