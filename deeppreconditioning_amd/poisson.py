"""Synthetic pressure-Poisson systems generated directly in HBM.

The reference cannot produce the 1M-DoF / 256^3 systems BASELINE.json names (its matrices come out
of OpenFOAM as dense lists, foam/newInterFoam/pEqn.H:54-68), so they are synthesised: closed-form
5-point / 7-point Laplacians, kron(I,T)+kron(T,I)[+...] with T = tridiag(-1,2,-1).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .operators import CsrSystem, _dev_ptr, _stream


def poisson_sizes(dim: int, n: int) -> tuple[int, int]:
    rows, nnz = C.c_int64(), C.c_int64()
    L.check(L.lib().dpcg_poisson_sizes(dim, n, C.byref(rows), C.byref(nnz)))
    return rows.value, nnz.value


def poisson_csr(dim: int, n: int, device=None, dtype=torch.float64):
    """(rowptr, col, val) CUDA tensors of the dim-D Poisson matrix on an n^dim grid."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    rows, nnz = poisson_sizes(dim, n)
    rowptr = torch.empty(rows + 1, dtype=torch.int32, device=device)
    col = torch.empty(nnz, dtype=torch.int32, device=device)
    val = torch.empty(nnz, dtype=dtype, device=device)
    with torch.cuda.device(device):
        L.check(L.lib().dpcg_gen_poisson(dim, n, _dev_ptr(rowptr), _dev_ptr(col), _dev_ptr(val),
                                         L.F64 if dtype == torch.float64 else L.F32, _stream()))
    return rowptr, col, val


def poisson_system(dim: int, n: int, device=None, dtype=torch.float64) -> CsrSystem:
    rowptr, col, val = poisson_csr(dim, n, device, dtype)
    return CsrSystem(rowptr, col, val, rowptr.numel() - 1)


def rhs(n: int, seed: int = 0, device=None) -> torch.Tensor:
    """b ~ U(-1,1) as generate_data.py:106, seeded like the golden fixtures (`default_rng(seed)`)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return torch.from_numpy(np.random.default_rng(seed).uniform(-1.0, 1.0, n)).to(device)
