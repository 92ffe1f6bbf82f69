"""ctypes binding of oracle/pcg_oracle.c -- TEST INFRASTRUCTURE ONLY (see that file's header)."""

from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = pathlib.Path(__file__).resolve().parent
_LIB_PATH = _HERE / "liboracle_pcg.so"

KINDS = {"none": 0, "jacobi": 1, "csr": 2, "llt_multiply": 3, "llt_solve": 4}


def build(force: bool = False) -> pathlib.Path:
    """Compile the C restatement with gcc (no GPU, no reference needed)."""
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < (_HERE / "pcg_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "liboracle_pcg.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.orc_pcg.restype = C.c_double
        _lib.orc_pcg_mixed.restype = C.c_double
        _lib.orc_pcg_perm.restype = C.c_double
        _lib.orc_dot.restype = C.c_double
        _lib.orc_ic0.restype = C.c_int64
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _csr_parts(M):
    if M is None:
        return None, None, None
    M = M.tocsr()
    return (np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
            np.ascontiguousarray(M.data, dtype=np.float64))


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(C.c_int(int(n)))


def spmv(A: sp.csr_matrix, x: np.ndarray) -> np.ndarray:
    rp, ci, v = _csr_parts(A)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(A.shape[0], dtype=np.float64)
    lib().orc_spmv(C.c_int64(A.shape[0]), _p(rp), _p(ci), _p(v), _p(x), _p(y))
    return y


def spmv_f32(A: sp.csr_matrix, x: np.ndarray) -> np.ndarray:
    rp, ci, _ = _csr_parts(A)
    v = np.ascontiguousarray(A.data, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty(A.shape[0], dtype=np.float32)
    lib().orc_spmv_f32(C.c_int64(A.shape[0]), _p(rp), _p(ci), _p(v), _p(x), _p(y))
    return y


def spmv_mixed(A: sp.csr_matrix, x: np.ndarray) -> np.ndarray:
    """fp64 y = fp64(fp32(A)) fp64(fp32(x)): fp32-STORED operands, fp64 products and in-order row sums (config 5's
    `A @ pk`).  `dpcg_spmv_f32` returns exactly fp32(y)."""
    rp, ci, _ = _csr_parts(A)
    v = np.ascontiguousarray(A.data, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty(A.shape[0], dtype=np.float64)
    lib().orc_spmv_mixed(C.c_int64(A.shape[0]), _p(rp), _p(ci), _p(v), _p(x), _p(y))
    return y


def dot(a: np.ndarray, b: np.ndarray) -> float:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return float(lib().orc_dot(C.c_int64(len(a)), _p(a), _p(b)))


def ic0(A: sp.csr_matrix) -> sp.csr_matrix:
    """IC(0) factor L of SPD A (pattern tril(A)); raises on a non-positive pivot."""
    T = sp.tril(A, format="csr")
    T.sort_indices()
    rp, ci, v = _csr_parts(T)
    v = v.copy()
    bad = lib().orc_ic0(C.c_int64(T.shape[0]), _p(rp), _p(ci), _p(v))
    if bad:
        raise ArithmeticError(f"IC(0) breakdown at row {bad - 1}")
    return sp.csr_matrix((v, ci, rp), shape=T.shape)


def transpose_csr(L: sp.csr_matrix) -> sp.csr_matrix:
    Lt = L.T.tocsr()
    Lt.sort_indices()
    return Lt


def sptrsv_lower(L: sp.csr_matrix, r: np.ndarray) -> np.ndarray:
    rp, ci, v = _csr_parts(L)
    r = np.ascontiguousarray(r, dtype=np.float64)
    y = np.empty_like(r)
    lib().orc_sptrsv_lower(C.c_int64(L.shape[0]), _p(rp), _p(ci), _p(v), _p(r), _p(y))
    return y


def sptrsv_upper(U: sp.csr_matrix, y: np.ndarray) -> np.ndarray:
    rp, ci, v = _csr_parts(U)
    y = np.ascontiguousarray(y, dtype=np.float64)
    z = np.empty_like(y)
    lib().orc_sptrsv_upper(C.c_int64(U.shape[0]), _p(rp), _p(ci), _p(v), _p(y), _p(z))
    return z


def spmv_vector(A: sp.csr_matrix, x: np.ndarray, tpr: int) -> np.ndarray:
    """y = A x with every row summed as the CSR-vector kernel sums it (`tpr` lanes per row, aligned pairs, shuffle tree:
    orc_spmv_vector) -- the bits of `S @ x` when reduction_geometry() says spmv_kernel "vector", spmv_tpr `tpr`."""
    rp, ci, v = _csr_parts(A)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(A.shape[0], dtype=np.float64)
    lib().orc_spmv_vector(C.c_int64(A.shape[0]), _p(rp), _p(ci), _p(v), _p(x), _p(y), C.c_int(int(tpr)))
    return y


def factor_levels(L: sp.csr_matrix) -> np.ndarray:
    """Level of every row of a lower-triangular factor (diagonal last in a row): 0 without off-diagonal entries, else 1 + the highest
    level among the rows it depends on -- the level sets the device schedules its triangular solves by."""
    n = L.shape[0]
    rp, ci = L.indptr.astype(np.int64), L.indices
    offd = np.ones(L.nnz, dtype=bool)
    offd[rp[1:] - 1] = False
    col = ci[offd]
    per_row = np.diff(rp) - 1
    has = per_row > 0
    starts = np.concatenate(([0], np.cumsum(per_row)))[:-1][has]
    level = np.zeros(n, dtype=np.int32)
    while True:
        new = np.zeros(n, dtype=np.int32)
        if col.size:
            new[has] = np.maximum.reduceat(level[col] + 1, starts)
        if np.array_equal(new, level):
            return level
        level = new


def sweep_rows(L: sp.csr_matrix, handle_index: np.ndarray) -> list:
    """The rows (as handle indices) every colour-sweep launch that adds to <r,z> walks, in its order: the levels of the UPPER solve are
    those of the lower one read backwards, a level's rows in ascending handle index (dpcg_precond.hip: compute_levels(order_by),
    reversed_levels).  handle_index[i]: the handle's index of factor row i.  For `device_tree["sweep_rows"]`."""
    level = factor_levels(L)
    nl = int(level.max()) + 1 if level.size else 0
    hi = np.asarray(handle_index)
    return [np.sort(hi[level == nl - 1 - k]).astype(np.int32) for k in range(nl)]


def pcg(A: sp.csr_matrix, b: np.ndarray, kind: str = "none", *, dinv=None, M=None, L=None, x0=None, rtol=1e-8,
        max_iter=1024, init_check="z", mixed=False, precond_perm=None, device_tree=None):
    """Returns (seconds, iterations, residual_history, x) -- same tuple as oracle.oracle's PCG.
    mixed=True: config 5, the loop's `A @ pk` on fp32-stored values and pk (orc_pcg_mixed).
    precond_perm: A (and b, x0, x) are the permuted system P A_c P^T, row i = the caller's row precond_perm[i], while
    dinv / M / L stay in the caller's numbering: z' = P M P^T r' (orc_pcg_perm).
    device_tree: {"spmv_grid", "nrb", "cyclic", "vec_grid"} (CsrSystem.reduction_geometry() of the handle under test): every dot
    product is summed in the DEVICE's reduction tree (orc_set_dot_tree) instead of the oracle's fixed blocks -- for kind "none" /
    "jacobi" the history then equals the multi-launch HIP solve's BIT FOR BIT.  "form": "small" (+ "small_threads") restates the
    one-workgroup solve of systems up to 6144 rows -- every preconditioner kind it serves: none / jacobi / csr / llt_multiply --,
    "form": "team" the 32-workgroup team solve, "form": "chip" (+ "rows_per_workgroup", "lanes_per_row" from chip_info()) the whole-chip solve.  "rz_kind" (+ "m_grid", "m_nrb", "m_cyclic"): who sums <r,z> behind an APPLIED
    preconditioner in the multi-launch form (orc_set_rz_tree) -- with it "csr", "llt_multiply" and "llt_solve" match bit for bit too,
    as long as the handle's answer is 0..4 (4: colour sweeps; then also "sweep_grid", "sweep_modes", "sweep_rows")."""
    if device_tree is not None:
        form = {"multi": 0, "small": 1, "team": 2, "chip": 3}[device_tree.get("form", "multi")]
        # ("chip": the whole-chip solve -- "rows_per_workgroup" from CsrSystem.chip_info() travels in the small_threads slot)
        lib().orc_set_dot_tree(1, int(device_tree["spmv_grid"]), int(device_tree["nrb"]),
                               int(device_tree.get("lanes_per_row", 1) if form == 3 else device_tree["cyclic"]),
                               int(device_tree["vec_grid"]), form,
                               int(device_tree["rows_per_workgroup"] if form == 3 else device_tree.get("small_threads", 0)))
        rzk = int(device_tree.get("rz_kind", 0))
        if form == 0 and rzk not in (0, 1, 2, 3, 4):
            raise ValueError("device_tree: the handle sums <r,z> in a tree this oracle does not restate (CSR-vector kernel, more than 16 sweeps)")
        lib().orc_set_rz_tree(rzk, int(device_tree.get("m_grid", 0)), int(device_tree.get("m_nrb", 0)), int(device_tree.get("m_cyclic", 0)))
        # CSR-vector kernels (rows of many entries): lanes per row for the system's matrix, for M (or L) and for L^T
        vec = (int(device_tree.get("spmv_tpr", 0)), int(device_tree.get("m_tpr", 0)), int(device_tree.get("mt_tpr", 0))) if form == 0 else (0, 0, 0)
        lib().orc_set_vector_tree(*[C.c_int(t) for t in vec])
        if form == 0 and rzk == 4:
            # colour sweeps: "sweep_grid", "sweep_modes" (one per launch) from reduction_geometry(), "sweep_rows": per launch the handle's
            # indices of the level's rows in level-major order (the caller derives them from the factor: tests/test_meshes.py::sweep_rows)
            rows = [np.ascontiguousarray(r, dtype=np.int32) for r in device_tree["sweep_rows"]]
            modes = np.ascontiguousarray(device_tree["sweep_modes"], dtype=np.int32)
            if len(rows) != len(modes):
                raise ValueError("device_tree: one row list per sweep launch")
            cnt = np.array([r.size for r in rows], dtype=np.int32)
            cat = np.ascontiguousarray(np.concatenate(rows) if rows else np.zeros(0, np.int32), dtype=np.int32)
            lib().orc_set_sweep_tree(C.c_int(len(rows)), C.c_int(int(device_tree["sweep_grid"])), _p(cnt), _p(modes), _p(cat))
        try:
            return pcg(A, b, kind, dinv=dinv, M=M, L=L, x0=x0, rtol=rtol, max_iter=max_iter, init_check=init_check, mixed=mixed,
                       precond_perm=precond_perm)
        finally:
            lib().orc_set_dot_tree(0, 0, 0, 0, 0, 0, 0)
            lib().orc_set_vector_tree(0, 0, 0)
    n = A.shape[0]
    rp, ci, v = _csr_parts(A)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x0a = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64)
    dinv_a = None if dinv is None else np.ascontiguousarray(dinv, dtype=np.float64)
    m = _csr_parts(M)
    l = _csr_parts(L)
    lt = _csr_parts(transpose_csr(L)) if L is not None else (None, None, None)
    x = np.empty(n, dtype=np.float64)
    hist = np.full(max_iter + 1, np.nan)
    iters = C.c_int(0)
    tail = (_p(b), _p(x0a), C.c_int(KINDS[kind]), _p(dinv_a),
            _p(m[0]), _p(m[1]), _p(m[2]), _p(l[0]), _p(l[1]), _p(l[2]), _p(lt[0]), _p(lt[1]), _p(lt[2]),
            C.c_double(rtol), C.c_int(max_iter), C.c_int(1 if init_check == "z" else 0), _p(x), _p(hist), C.byref(iters))
    if precond_perm is not None:
        pp = np.ascontiguousarray(precond_perm, dtype=np.int32)
        v32 = v.astype(np.float32) if mixed else None
        sec = lib().orc_pcg_perm(C.c_int64(n), _p(rp), _p(ci), _p(v), _p(v32), _p(pp), *tail)
    elif mixed:
        v32 = v.astype(np.float32)
        sec = lib().orc_pcg_mixed(C.c_int64(n), _p(rp), _p(ci), _p(v), _p(v32), *tail)
    else:
        sec = lib().orc_pcg(C.c_int64(n), _p(rp), _p(ci), _p(v), *tail)
    k = iters.value
    return float(sec), k, hist[: k + 1].copy(), x
