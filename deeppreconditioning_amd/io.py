"""Readers for the reference's on-disk formats and the device COO -> CSR step that feeds them to the solver
(SURVEY.md 8-f3).  File parsing is host plumbing (numpy); the triplets are sorted, de-duplicated and compressed to
CSR on the GPU by `dpcg_coo_to_csr`.

Formats
  matrix.npz             scipy COO written by `save_npz(..., compressed=False)` (generate_data.py:109); the data sets
                         unpack it positionally as `rows, columns, _, original_size, values` (data_set.py:85)
  right_hand_side.csv,   `np.savetxt`, one value per line (generate_data.py:110-111, data_set.py:99-100)
  solution.csv
  matrix.csv             OpenFOAM dump `i,j,%.32f`, every non-zero of the full symmetric matrix (pEqn.H:98-108);
                         the sign is flipped on load (generate_data.py:67-71)
  StAn *.npz             `indices (2,nnz), values, solution, rhs` in that order (data_set.py:186-188)
"""

from __future__ import annotations

import ctypes as C
import pathlib

import numpy as np
import torch

from . import _lib as L
from .operators import CsrSystem, _dev_ptr, _stream


def coo_to_csr_device(rows, cols, vals, n: int, device=None):
    """(rowptr int32[n+1], col int32[nnz'], val float64[nnz']) CUDA tensors from coordinate triplets; duplicates are
    summed, columns ascend within a row."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    r = torch.as_tensor(np.asarray(rows) if not isinstance(rows, torch.Tensor) else rows).to(device=device, dtype=torch.int32).contiguous()
    c = torch.as_tensor(np.asarray(cols) if not isinstance(cols, torch.Tensor) else cols).to(device=device, dtype=torch.int32).contiguous()
    v = torch.as_tensor(np.asarray(vals) if not isinstance(vals, torch.Tensor) else vals).to(device=device, dtype=torch.float64).contiguous()
    nnz = r.numel()
    if c.numel() != nnz or v.numel() != nnz:
        raise ValueError("rows, cols, vals must have the same length")
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=device)
    col = torch.empty(max(nnz, 1), dtype=torch.int32, device=device)
    val = torch.empty(max(nnz, 1), dtype=torch.float64, device=device)
    out = C.c_int64()
    with torch.cuda.device(device):
        L.check(L.lib().dpcg_coo_to_csr(n, nnz, _dev_ptr(r), _dev_ptr(c), _dev_ptr(v), _dev_ptr(rowptr), _dev_ptr(col),
                                        _dev_ptr(val), C.byref(out), _stream()))
    return rowptr, col[: out.value].contiguous(), val[: out.value].contiguous()


def system_from_coo(rows, cols, vals, n: int, device=None) -> CsrSystem:
    rowptr, col, val = coo_to_csr_device(rows, cols, vals, n, device)
    return CsrSystem(rowptr, col, val, n)


# ---- file parsers (host) -------------------------------------------------------------------------
def load_matrix_npz(path):
    """scipy COO npz -> (rows, cols, vals, n).  Positional unpacking as data_set.py:85."""
    with np.load(path) as f:
        rows, cols, _fmt, shape, vals = (f[k] for k in f.files)
    return rows.astype(np.int64), cols.astype(np.int64), vals.astype(np.float64), int(shape[0])


def load_openfoam_matrix_csv(path):
    """OpenFOAM `matrix.csv` (pEqn.H:98-108) -> (rows, cols, vals, n) with the sign flip of generate_data.py:71."""
    data = np.genfromtxt(path, delimiter=",")
    data = data.reshape(-1, 3)
    rows, cols = data[:, 0].astype(np.int64), data[:, 1].astype(np.int64)
    vals = -data[:, 2]
    return rows, cols, vals, int(rows.max()) + 1


def load_stan_npz(path):
    """StAn sample -> (rows, cols, vals, n, solution, rhs) (data_set.py:186-188)."""
    with np.load(path) as f:
        indices, values, solution, rhs = (f[k] for k in f.files)
    return indices[0].astype(np.int64), indices[1].astype(np.int64), values.astype(np.float64), len(solution), solution, rhs


def load_vector(path) -> np.ndarray:
    return np.atleast_1d(np.loadtxt(path)).astype(np.float64)


def load_case(folder, device=None):
    """One `case_XXXX/` directory of the sludge-pattern data set (generate_data.py:97-111):
    returns (CsrSystem, right_hand_side, solution) with the vectors on the system's device."""
    folder = pathlib.Path(folder)
    rows, cols, vals, n = load_matrix_npz(folder / "matrix.npz")
    system = system_from_coo(rows, cols, vals, n, device)
    b = torch.from_numpy(load_vector(folder / "right_hand_side.csv")).to(system.device)
    x = torch.from_numpy(load_vector(folder / "solution.csv")).to(system.device)
    return system, b, x


def write_case(matrix_csv, case_directory, rng=None, device=None) -> dict:
    """The post-simulation half of generate_data.py:97-111 for one OpenFOAM dump: build the matrix (sign flip,
    symmetry check, generate_data.py:55-81 -- the dense eigenvalue check is replaced by CG's own breakdown test), draw
    the right-hand side from U(-1, 1), solve for the ground truth with the absolute criterion of
    `scipy.sparse.linalg.cg(rtol=0, atol=1e-6)` (||r|| <= 1e-6, generate_data.py:107) on the GPU, and write
    `matrix.npz` / `right_hand_side.csv` / `solution.csv` the way the data sets read them."""
    import scipy.sparse as sp
    rng = np.random.default_rng(69) if rng is None else rng
    rows, cols, vals, n = load_openfoam_matrix_csv(matrix_csv)
    matrix = sp.coo_matrix((vals, (rows, cols)), shape=(n, n))
    assert (matrix.transpose() != matrix).nnz == 0, "Generated matrix is non-symmetric matrix"      # generate_data.py:76
    system = system_from_coo(rows, cols, vals, n, device)
    right_hand_side = rng.uniform(-1, 1, size=n)
    result = system.solve(torch.from_numpy(right_hand_side).to(system.device), rtol_sq=0.0, atol_sq=1e-12,
                          max_iter=10 * n, want_history=False)
    assert result.status == 0, "Generated matrix is not positive definite (CG did not converge)"
    case_directory = pathlib.Path(case_directory)
    case_directory.mkdir(parents=True, exist_ok=True)
    sp.save_npz(case_directory / "matrix.npz", matrix, compressed=False)                             # generate_data.py:109
    np.savetxt(case_directory / "right_hand_side.csv", right_hand_side)
    np.savetxt(case_directory / "solution.csv", result.x.cpu().numpy())
    system.close()
    return {"n": n, "iterations": result.iterations, "final_res": result.final_res}
