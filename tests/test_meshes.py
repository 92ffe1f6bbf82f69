"""BASELINE config 3 on genuinely unstructured matrices: a quadtree-refined finite-volume Laplacian (hanging nodes, holes,
hexRef8 numbering) and a Delaunay graph Laplacian (deeppreconditioning_amd/meshes.py, restated in oracle/oracle.py).

CPU part: the product's generators and the oracle's give the same bits; the matrices are what they claim to be (symmetric,
positive diagonal / non-positive off-diagonals, irreducibly diagonally dominant, irregular degrees, triangles).
GPU part (`-m gpu`): SpMV bit-exact; Jacobi and IC(0) in the caller's order against oracle/pcg_oracle.c with the device's reduction
trees -- counts, histories and x EQUAL --; IC(0) in multicolour order (colour sweeps) counts equal and histories within north_star's
1e-10 while the recurrence is stable -- at ~10K rows and at the ~1M rows config 3 names.
"""

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import c_oracle as CO
from oracle import oracle as O

HIST_RTOL = 1e-10


def _product_meshes():
    import importlib.util
    import pathlib
    path = pathlib.Path(__file__).resolve().parent.parent / "deeppreconditioning_amd" / "meshes.py"
    spec = importlib.util.spec_from_file_location("dpcg_meshes_under_test", path)       # (no torch / GPU needed for this file)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _same(A, B):
    return (A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
            and np.array_equal(A.data, B.data))


def test_mesh_generators_product_and_oracle_agree():
    M = _product_meshes()
    for numbering in ("foam", "random"):
        assert _same(M.quadtree_fv_laplacian(72, 3, numbering=numbering), O.quadtree_fv_laplacian(72, 3, numbering=numbering))
    assert _same(M.delaunay_laplacian(5000, 2), O.delaunay_laplacian(5000, 2))
    assert not _same(O.quadtree_fv_laplacian(72, 3), O.quadtree_fv_laplacian(72, 4))


@pytest.mark.parametrize("name,make", [("quadtree_foam", lambda: O.quadtree_fv_laplacian(80, 1)),
                                       ("quadtree_random", lambda: O.quadtree_fv_laplacian(80, 1, numbering="random")),
                                       ("delaunay", lambda: O.delaunay_laplacian(6000, 1))])
def test_mesh_matrices_are_spd_m_matrices_with_irregular_structure(name, make):
    A = make()
    n = A.shape[0]
    assert A.indices.dtype == np.int32 and A.has_sorted_indices and abs(A - A.T).max() == 0.0
    d = A.diagonal()
    off = A - sp.diags(d)
    assert d.min() > 0 and off.data.max() < 0                                  # generate_data.py:71: positive diagonal, negative couplings
    slack = d + np.asarray(off.sum(axis=1)).ravel()
    assert slack.min() > -1e-12 * d.max() and (slack > 1e-9).sum() > 0          # weakly dominant everywhere, strictly on the boundary
    assert sp.csgraph.connected_components(A, directed=False)[0] == 1           # irreducible -> positive definite
    assert np.linalg.eigvalsh(A.toarray()).min() > 0 if n <= 7000 else True
    deg = np.diff(A.indptr)
    assert len(np.unique(deg)) >= 4                                             # irregular degree
    T = (off != 0).astype(np.int32)
    assert (T @ T).multiply(T).sum() > 0                                        # triangles in the graph
    if name == "quadtree_foam":
        r = np.repeat(np.arange(n), deg)
        assert abs(r - A.indices).max() > n // 2                                # appended children: couplings across the whole numbering
    # the oracle solves it: Jacobi PCG converges, IC(0) exists and preconditions better
    b = O.rhs(n, 0)
    _, it_j, hist, x = CO.pcg(A, b, "jacobi", dinv=O.jacobi_dinv(A))
    assert hist[-1] < 1e-8 and np.linalg.norm(b - A @ x) / np.linalg.norm(b) < 2e-4
    _, it_c, _, _ = CO.pcg(A, b, "llt_solve", L=CO.ic0(A))
    assert it_c < it_j


# ---- GPU ---------------------------------------------------------------------------------------------------------------
def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _permuted(A, perm):
    B = A[perm][:, perm].tocsr()
    B.sort_indices()
    return B


@pytest.fixture(scope="module")
def D():
    import torch
    import deeppreconditioning_amd as pkg
    assert torch.cuda.is_available(), "these tests need the GPU"
    pkg._lib.lib()
    return pkg


def _oracle_pair(B, b, kind, **kw):
    """oracle/pcg_oracle.c twice: on b, and on b with every entry moved by at most one unit in the last place.  The drift
    between the two histories is the system's own amplification of a rounding-level perturbation: IC(0)-PCG on these
    meshes is sensitive (the numpy and the C oracle -- same algorithm, other summation order inside the triangular
    solves -- agree to 1e-13 at first and to 1e-6 .. 3e-2 after ~80 updates on the 10K-row quadtree systems)."""
    _, it, hist, x = CO.pcg(B, b, kind, **kw)
    eps = np.random.default_rng(99).integers(-1, 2, b.shape[0]) * 2.0 ** -52
    _, it_p, hist_p, _ = CO.pcg(B, b * (1.0 + eps), kind, **kw)
    m = min(len(hist), len(hist_p))
    drift = np.zeros(len(hist))
    drift[:m] = np.abs(hist[:m] - hist_p[:m]) / hist[:m]
    drift[m:] = np.inf
    return it, hist, x, abs(it - it_p), np.maximum.accumulate(drift)


def _check_history(res, it, hist, dcount, drift, tag):
    """Counts equal and histories within north_star's 1e-10 wherever the recurrence is stable; where a one-ulp perturbation
    of b already moves the ORACLE's history by more than 1e-12, the bound is 100 x that drift (and the count may differ by
    what the perturbation itself changes, + 1)."""
    assert abs(res.iterations - it) <= (0 if dcount == 0 and drift.max() < 1e-12 else dcount + 1), (tag, res.iterations, it)
    m = min(len(hist), len(res.res_history))
    tol = np.maximum(HIST_RTOL, 100.0 * drift[:m])
    err = np.abs(res.res_history[:m] - hist[:m]) / hist[:m]
    assert np.all(err <= tol), (tag, int(np.argmax(err > tol)), float(err.max()), float(drift.max()))
    stable = int(np.argmax(drift > 1e-12)) if (drift > 1e-12).any() else m
    assert stable >= min(m, 20), (tag, stable)              # the tight bar covers a real stretch of every history


def _mesh_parity(D, A, expect, max_iter, reorder="auto"):
    n = A.shape[0]
    b = O.rhs(n, 0)
    S = D.CsrSystem.from_any(A, reorder=reorder)
    info = S.info()
    if expect.get("reordered") is not None:
        assert S.reordered == expect["reordered"], info
    if expect.get("kernel"):
        assert info["spmv_kernel"] in expect["kernel"], info
    perm = S.permutation() if S.reordered else np.arange(n)
    B = _permuted(A, perm) if S.reordered else A
    x = O.rhs(n, 7)
    y = (S @ _dev(x)).cpu().numpy()
    assert np.array_equal(y[perm], CO.spmv(B, x[perm]))                       # bit-exact on the matrix the handle iterates on
    # Jacobi
    S.set_preconditioner(D.Jacobi())
    NO_SMALL = D._lib.NO_SMALL
    res = S.solve(_dev(b), max_iter=max_iter, flags=NO_SMALL)
    # (round 4) the oracle adds its dot products in the device's reduction tree: history, count and x are EQUAL
    _, it, hist, xs = CO.pcg(B, b[perm], "jacobi", dinv=O.jacobi_dinv(B), max_iter=max_iter, device_tree=S.reduction_geometry())
    assert res.iterations == it and np.array_equal(res.res_history, hist) and np.array_equal(res.x.cpu().numpy()[perm], xs)
    out = {"jacobi": it}
    # IC(0) in the caller's order: the factor of the CALLER's matrix bit for bit, applied by triangular solves
    S.set_preconditioner(D.IC0("solve"))
    Lref = CO.ic0(A)
    rp, ci, v = S.factor()
    assert np.array_equal(rp, Lref.indptr) and np.array_equal(ci, Lref.indices) and np.array_equal(v, Lref.data)
    zref = CO.sptrsv_upper(CO.transpose_csr(Lref), CO.sptrsv_lower(Lref, b))
    assert np.array_equal(S.precond_apply(_dev(b)).cpu().numpy(), zref)
    res = S.solve(_dev(b), max_iter=max_iter, flags=NO_SMALL)
    kw = dict(precond_perm=perm) if S.reordered else {}
    geo = S.reduction_geometry()
    assert geo["rz_kind"] in (1, 2), geo
    _, it, hist, _ = CO.pcg(B, b[perm], "llt_solve", L=Lref, max_iter=max_iter, device_tree=geo, **kw)
    assert res.iterations == it and np.array_equal(res.res_history, hist)      # bit for bit, rounding-sensitive as these systems are
    out["ic0_caller"] = (it, S.info()["levels_lower"])
    # IC(0) in multicolour order: IC(0) of Q A Q^T bit for bit, PCG through orc_pcg_perm
    S.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    nc, q = S.precond_ordering()
    assert np.array_equal(np.sort(q), np.arange(n)) and 2 <= nc <= expect.get("max_colors", 12)
    Lq = CO.ic0(_permuted(A, q))
    rp, ci, v = S.factor()
    assert np.array_equal(rp, Lq.indptr) and np.array_equal(ci, Lq.indices) and np.array_equal(v, Lq.data)
    assert S.info()["levels_lower"] <= nc
    zq = np.empty(n)
    zq[q] = CO.sptrsv_upper(CO.transpose_csr(Lq), CO.sptrsv_lower(Lq, b[q]))
    assert np.array_equal(S.precond_apply(_dev(b)).cpu().numpy(), zq)
    qinv = np.empty(n, dtype=np.int32)
    qinv[q] = np.arange(n, dtype=np.int32)
    res = S.solve(_dev(b), max_iter=max_iter, flags=NO_SMALL)
    geo = S.reduction_geometry()
    if geo["rz_kind"] == 4:      # colour sweeps: <r,z> summed launch by launch over the levels -- restated too (orc_set_sweep_tree)
        iperm = np.empty(n, dtype=np.int64)
        iperm[perm] = np.arange(n)
        geo["sweep_rows"] = CO.sweep_rows(Lq, iperm[q])
        assert len(geo["sweep_rows"]) == len(geo["sweep_modes"]) == S.info()["levels_upper"]
    if geo["rz_kind"] in (1, 2, 4):
        _, it, hist, _ = CO.pcg(B, b[perm], "llt_solve", L=Lq, precond_perm=qinv[perm], max_iter=max_iter, device_tree=geo)
        assert res.iterations == it and np.array_equal(res.res_history, hist)
    else:        # a tree the checker does not restate: 1e-10 while the recurrence is stable (see _check_history)
        it, hist, _, dc, drift = _oracle_pair(B, b[perm], "llt_solve", L=Lq, precond_perm=qinv[perm], max_iter=max_iter)
        _check_history(res, it, hist, dc, drift, "ic0 multicolour")
    out["ic0_multicolor"] = (it, nc)
    out["ic0_multicolor_rz_kind"] = geo["rz_kind"]
    S.close()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", [("quadtree_foam", lambda: O.quadtree_fv_laplacian(96, 2)),
                                       ("quadtree_random", lambda: O.quadtree_fv_laplacian(96, 2, numbering="random")),
                                       ("delaunay", lambda: O.delaunay_laplacian(12000, 2))])
def test_unstructured_meshes_small(D, name, make):
    """The launch forms below the 1M-DoF kernels (two-kernel updates, gather SpMV, LDS-ring / sync-free triangular solves) on the
    same kinds of matrices."""
    out = _mesh_parity(D, make(), {}, 1024)
    assert out["ic0_caller"][0] < out["jacobi"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,expect", [
    ("quadtree_foam_1M", lambda: O.quadtree_fv_laplacian(1000, 0), {"reordered": True, "kernel": ("tile",), "max_colors": 8}),   # (by regions)
    ("quadtree_random_1M", lambda: O.quadtree_fv_laplacian(1000, 0, numbering="random"),
     {"reordered": True, "kernel": ("tile",), "max_colors": 8}),
    ("delaunay_1M", lambda: O.delaunay_laplacian(1000000, 0), {"reordered": True, "kernel": ("tile",), "max_colors": 9})])
def test_config3_unstructured_meshes_million_dof(D, name, make, expect):
    """BASELINE config 3 ("OpenFOAM interFoam pressure-correction matrix, ~1M DoF, unstructured CSR") on matrices with irregular
    degree, triangles and no grid structure: the plain call (the library reorders a scattered numbering by itself -- reverse
    Cuthill-McKee; OpenFOAM's numbering, fine on average but too scattered in 12 % of its row blocks, region by region -- and plans
    the x-tile SpMV on the result), 60 updates each of Jacobi, IC(0) in the caller's order (level-scheduled L / L^T solves) and IC(0)
    in multicolour order against the C oracle on the system the handle iterates on, the oracle's dot products in the device's
    reduction trees -- the colour sweeps' launch-by-launch <r,z> included: counts and histories EQUAL."""
    out = _mesh_parity(D, make(), expect, 60)
    assert out["ic0_multicolor_rz_kind"] == 4          # 4-5 wide levels: colour sweeps, restated (never the tolerance branch)


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", [("quadtree_foam_1M", lambda: O.quadtree_fv_laplacian(1000, 0)),
                                       ("quadtree_random_1M", lambda: O.quadtree_fv_laplacian(1000, 0, numbering="random")),
                                       ("delaunay_1M", lambda: O.delaunay_laplacian(1000000, 0))])
def test_million_row_meshes_full_length_histories(D, name, make):
    """The 1M-row meshes to the END of their histories (the reference's defaults: rtol 1e-8 squared, cap 1024 -- cg.py:51), not the
    first 60 updates: Jacobi and IC(0) in multicolour order through the plain multi-launch path against the C oracle with the device's
    reduction trees on the system the handle iterates on -- counts, every residual of up to 1025, and x under Jacobi EQUAL; Jacobi also
    through the DEFAULT call (the streamed whole-chip form) and, on one mesh, with config 5's fp32-stored operands, to the end."""
    A = make()
    n = A.shape[0]
    b = O.rhs(n, 0)
    S = D.CsrSystem.from_any(A)
    perm = S.permutation() if S.reordered else np.arange(n)
    B = _permuted(A, perm) if S.reordered else A
    NO_SMALL = D._lib.NO_SMALL
    S.set_preconditioner(D.Jacobi())
    res = S.solve(_dev(b), flags=NO_SMALL)
    _, it, hist, xs = CO.pcg(B, b[perm], "jacobi", dinv=O.jacobi_dinv(B), device_tree=S.reduction_geometry())
    assert res.iterations == it and len(hist) == it + 1 and np.array_equal(res.res_history, hist), (name, res.iterations, it)
    assert np.array_equal(res.x.cpu().numpy()[perm], xs)
    # ... and the DEFAULT call (the one-launch kernel with the matrix streamed, k_pcg_chip MODE 5) to the end of ITS history, x included,
    # against the oracle with the whole-chip reduction tree; then config 5's fp32-stored operands (MODE 6) likewise
    ci = S.chip_info()
    assert ci["chip_by_default"], ci
    chip_tree = {**S.reduction_geometry(), "form": "chip", "rows_per_workgroup": ci["rows_per_workgroup"]}
    res_d = S.solve(_dev(b))
    _, it_d, hist_d, xs_d = CO.pcg(B, b[perm], "jacobi", dinv=O.jacobi_dinv(B), device_tree=chip_tree)
    assert res_d.iterations == it_d and len(hist_d) == it_d + 1 and np.array_equal(res_d.res_history, hist_d), (name, res_d.iterations, it_d)
    assert np.array_equal(res_d.x.cpu().numpy()[perm], xs_d)
    assert not np.array_equal(res_d.res_history, res.res_history)          # (another summation order: it WAS the other form)
    if name in ("quadtree_random_1M", "quadtree_foam_1M"):      # (RCM order: 4-byte value slots, resident -- MODE 4; OpenFOAM's numbering: no band, streamed -- MODE 6)
        res_m = S.solve(_dev(b), flags=D._lib.SPMV_F32)
        _, it_m, hist_m, xs_m = CO.pcg(B, b[perm], "jacobi", dinv=O.jacobi_dinv(B), mixed=True, device_tree=chip_tree)
        assert res_m.iterations == it_m and np.array_equal(res_m.res_history, hist_m), (name, res_m.iterations, it_m)
        assert np.array_equal(res_m.x.cpu().numpy()[perm], xs_m)
    S.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    nc, q = S.precond_ordering()
    Lq = CO.ic0(_permuted(A, q))
    qinv = np.empty(n, dtype=np.int32)
    qinv[q] = np.arange(n, dtype=np.int32)
    res = S.solve(_dev(b), flags=NO_SMALL)
    geo = S.reduction_geometry()
    assert geo["rz_kind"] == 4, geo
    iperm = np.empty(n, dtype=np.int64)
    iperm[perm] = np.arange(n)
    geo["sweep_rows"] = CO.sweep_rows(Lq, iperm[q])
    _, it_c, hist_c, _ = CO.pcg(B, b[perm], "llt_solve", L=Lq, precond_perm=qinv[perm], device_tree=geo)
    assert res.iterations == it_c and np.array_equal(res.res_history, hist_c), (name, res.iterations, it_c)
    assert it_c <= it            # (IC(0) never needs more updates than Jacobi here; the Delaunay system converges well inside the cap)
    S.close()


def _chunks_per_block(B):
    """chunks of 64 x entries that the columns of each 256-row block touch (what the x-tile plan stages: at most 40)"""
    n = B.shape[0]
    blk = np.repeat(np.arange(n), np.diff(B.indptr)) // 256
    u = np.unique(blk.astype(np.int64) * (n // 64 + 2) + B.indices // 64)
    return np.bincount(u // (n // 64 + 2), minlength=(n + 255) // 256)


@pytest.mark.gpu
def test_region_by_region_numbering(D):
    """The cheap locality order for OpenFOAM's numbering (refinement appends cells: the x-gather is fine on average, 12 % of the row
    blocks are too scattered for the x-tile plan): forced on a 14K-row mesh -- a permutation, the same from create to create, PCG
    parity on the system the handle iterates on through `_mesh_parity` -- and chosen BY ITSELF for the 1M-row quadtree mesh in
    OpenFOAM's numbering, where the x-tile plan then takes every block.  A scattered numbering still goes through reverse
    Cuthill-McKee (whose band the one-launch solve needs), a banded one is left alone."""
    A = O.quadtree_fv_laplacian(96, 2)
    n = A.shape[0]
    S = D.CsrSystem.from_any(A, reorder="regions")
    perm = S.permutation()
    assert S.reordered and np.array_equal(np.sort(perm), np.arange(n))
    S2 = D.CsrSystem.from_any(A, reorder="regions")
    assert np.array_equal(S2.permutation(), perm)
    S2.close()
    S.close()
    _mesh_parity(D, A, {"reordered": True}, 200, reorder="regions")
    for tiny in (O.poisson2d(20), O.poisson2d(2), sp.csr_matrix(np.array([[2.0]]))):      # (fewer regions for few rows; one row)
        St = D.CsrSystem.from_any(tiny, reorder="regions")
        assert St.reordered and np.array_equal(np.sort(St.permutation()), np.arange(tiny.shape[0]))
        St.close()
    # 1M rows, OpenFOAM's numbering: AUTO takes the regions
    A = O.quadtree_fv_laplacian(1000, 0)
    n = A.shape[0]
    before = _chunks_per_block(A)
    assert (before > 40).mean() > 0.05
    S = D.CsrSystem.from_any(A)
    info = S.info()
    assert S.reordered and info["spmv_kernel"] == "tile" and info["gather_ratio"] < 4.0, info
    perm = S.permutation()
    assert np.array_equal(np.sort(perm), np.arange(n))
    B = _permuted(A, perm)
    after = _chunks_per_block(B)
    assert after.max() <= 40 and after.mean() < 0.7 * before.mean(), (after.max(), after.mean(), before.mean())
    x = O.rhs(n, 3)
    assert np.array_equal((S @ _dev(x)).cpu().numpy()[perm], CO.spmv(B, x[perm]))
    S.close()
    # a banded system of short rows keeps its numbering (the whole-chip solve needs the band), whatever its size
    S = D.CsrSystem.from_any(O.poisson2d(700))
    assert not S.reordered
    S.close()
