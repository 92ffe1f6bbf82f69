"""GPU operators behind the reference's duck-typed `A @ v` / `M @ v` protocol (cg.py:60,61,75,81).

`CsrSystem` owns a libdpcg handle for the system matrix A; the preconditioner classes describe how
`M @ r` is applied (test.py:70-105).  All arithmetic happens in the hand-written HIP kernels of
libdpcg.so; torch is used for device memory and streams only.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib as L

try:  # scipy is optional plumbing for host-side CSR inputs
    import scipy.sparse as _sp
except Exception:  # pragma: no cover
    _sp = None


# --------------------------------------------------------------------------------------------
# input normalisation (host logic, no GPU needed)
# --------------------------------------------------------------------------------------------
def _is_scipy(A) -> bool:
    return _sp is not None and _sp.issparse(A)


def csr_arrays(A):
    """Normalise a matrix-like to CSR parts: (space, rowptr, col, val, n).

    space = "host": numpy int32/int32/float64 arrays.  space = "device": torch CUDA tensors.
    Accepts scipy sparse, numpy 2-D, torch sparse-CSR / sparse-COO / dense tensors (the reference's
    callers pass DENSE fp64 tensors, test.py:61-68, train.py:93-95).  Columns ascend within a row.
    """
    if isinstance(A, tuple) and len(A) == 3:  # ready-made CSR parts (rowptr, col, val)
        rp, ci, v = A
        if isinstance(v, torch.Tensor):
            if v.is_cuda:
                return ("device", rp.to(torch.int32).contiguous(), ci.to(torch.int32).contiguous(),
                        v.to(torch.float64).contiguous(), rp.numel() - 1)
            rp, ci, v = rp.numpy(), ci.numpy(), v.numpy()
        return ("host", np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(ci, dtype=np.int32),
                np.ascontiguousarray(v, dtype=np.float64), len(rp) - 1)
    if _is_scipy(A):
        M = A.tocsr()
        if not M.has_canonical_format:
            M = M.copy()
            M.sum_duplicates()
        return ("host", np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
                np.ascontiguousarray(M.data, dtype=np.float64), M.shape[0])
    if isinstance(A, np.ndarray):
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("expected a square 2-D array")
        return csr_arrays(_sp.csr_matrix(A))
    if isinstance(A, torch.Tensor):
        if A.dim() != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("expected a square 2-D tensor")
        if A.layout != torch.sparse_csr:
            A = A.to_sparse_csr() if A.layout == torch.strided else A.coalesce().to_sparse_csr()
        rp, ci, v = A.crow_indices(), A.col_indices(), A.values()
        if A.is_cuda:
            return ("device", rp.to(torch.int32).contiguous(), ci.to(torch.int32).contiguous(),
                    v.to(torch.float64).contiguous(), A.shape[0])
        return ("host", np.ascontiguousarray(rp.numpy(), dtype=np.int32), np.ascontiguousarray(ci.numpy(), dtype=np.int32),
                np.ascontiguousarray(v.to(torch.float64).numpy()), A.shape[0])
    raise TypeError(f"cannot interpret {type(A).__name__} as a sparse system matrix")


def diagonal_only(rowptr, col, n) -> bool:
    """True when the CSR pattern is exactly one entry (i, i) per row -- M = diag(d) (test.py:74-79)."""
    if isinstance(rowptr, torch.Tensor):
        if rowptr.numel() != n + 1 or col.numel() != n:
            return False
        ar = torch.arange(n + 1, device=rowptr.device, dtype=rowptr.dtype)
        return bool(torch.equal(rowptr, ar) and torch.equal(col, ar[:-1]))
    return len(col) == n and np.array_equal(rowptr, np.arange(n + 1)) and np.array_equal(col, np.arange(n))


def _dev_ptr(t: torch.Tensor | None, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("expected a CUDA (ROCm) tensor")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


def _np_ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _to_device_f64(v, device) -> torch.Tensor:
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(np.ascontiguousarray(v))
    return v.to(device=device, dtype=torch.float64).contiguous()


# --------------------------------------------------------------------------------------------
# preconditioner descriptions (how `M @ r` is applied)
# --------------------------------------------------------------------------------------------
class Preconditioner:
    """Base: a description that `CsrSystem.set_preconditioner` turns into libdpcg state.

    Instances also work stand-alone as duck-typed operators (`M @ r`), so they drop into an
    unmodified copy of the reference loop; that path builds a private handle on first use.
    """

    _private: "CsrSystem | None" = None

    def _attach(self, system: "CsrSystem") -> None:
        raise NotImplementedError

    def _private_system(self) -> "CsrSystem":
        raise NotImplementedError

    def __matmul__(self, r: torch.Tensor) -> torch.Tensor:
        if self._private is None:
            self._private = self._private_system()
            self._attach(self._private)
        return self._private.precond_apply(r)


class Identity(Preconditioner):
    """M = I, the `vanilla` technique (test.py:70-72)."""

    def _attach(self, system):
        L.check(L.lib().dpcg_set_precond_none(system._h))

    def __matmul__(self, r):
        return r.clone()


class Jacobi(Preconditioner):
    """M = diag(1/a_ii) (test.py:74-79).  `dinv=None` extracts the diagonal of A on the device."""

    def __init__(self, dinv=None):
        self.dinv = dinv

    def _attach(self, system):
        if self.dinv is None:
            L.check(L.lib().dpcg_set_precond_jacobi(system._h, None, L.DEVICE, _stream()))
        else:
            d = _to_device_f64(self.dinv, system.device)
            if d.numel() != system.n:
                raise ValueError("dinv has the wrong length")
            L.check(L.lib().dpcg_set_precond_jacobi(system._h, _dev_ptr(d), L.DEVICE, _stream()))

    def _private_system(self):
        if self.dinv is None:
            raise ValueError("a stand-alone Jacobi operator needs explicit dinv")
        d = _to_device_f64(self.dinv, torch.device("cuda", torch.cuda.current_device()))
        n = d.numel()
        ar = torch.arange(n + 1, device=d.device, dtype=torch.int32)
        return CsrSystem(ar, ar[:-1].clone(), torch.ones(n, device=d.device, dtype=torch.float64), n)


class _CsrBacked(Preconditioner):
    def __init__(self, matrix):
        self.space, self.rowptr, self.col, self.val, self.n = csr_arrays(matrix)

    def _parts(self):
        if self.space == "host":
            return L.HOST, _np_ptr(self.rowptr), _np_ptr(self.col), _np_ptr(self.val), len(self.col)
        return L.DEVICE, _dev_ptr(self.rowptr), _dev_ptr(self.col), _dev_ptr(self.val), self.col.numel()

    def _private_system(self):
        if self.space == "host":
            return CsrSystem.from_host(self.rowptr, self.col, self.val, self.n)
        return CsrSystem(self.rowptr, self.col, self.val, self.n)


class CsrPreconditioner(_CsrBacked):
    """z = M r with M an explicit CSR matrix -- what test.py:88,105 hand to the solver (M = L L^T)."""

    def _attach(self, system):
        if self.n != system.n:
            raise ValueError("preconditioner size mismatch")
        space, rp, ci, v, nnz = self._parts()
        L.check(L.lib().dpcg_set_precond_csr(system._h, nnz, rp, ci, v, space, _stream()))


class LLtMultiply(_CsrBacked):
    """z = L (L^T r): the learned preconditioner of test.py:100-105 without forming L L^T."""

    mode = L.PRECOND_LLT_MULTIPLY

    def __init__(self, lower):
        super().__init__(lower)

    def _attach(self, system):
        if self.n != system.n:
            raise ValueError("factor size mismatch")
        space, rp, ci, v, nnz = self._parts()
        L.check(L.lib().dpcg_set_precond_llt(system._h, self.mode, nnz, rp, ci, v, space, _stream()))


class LLtSolve(LLtMultiply):
    """z = L^-T (L^-1 r): a true incomplete-Cholesky apply by level-scheduled triangular solves.

    The parallelism of a triangular solve is the width of the factor's dependency levels (`CsrSystem.info()`:
    `levels_lower` / `levels_upper`).  The factor of a grid has n^(1/2) .. n^(2/3)-row levels; a factor whose every row
    reaches back to its predecessor -- the sparsity the reference's CNN emits: 15 entries per row with column row-1 among
    them -- is ONE chain of n levels and is solved row after row (256^2: 65 536 levels, ~0.1 s per apply).  That factor is
    meant to be multiplied (`LLtMultiply`, the reference's own use, test.py:100-105)."""

    mode = L.PRECOND_LLT_SOLVE


class IC0(Preconditioner):
    """IC(0) of A computed at setup (stands in for ilupp.ichol0, test.py:83).

    mode="solve" applies it by triangular solves; mode="multiply" reproduces the reference's
    `_construct_incomplete_cholesky`, which multiplies by L L^T (test.py:88, marked unstable at
    test.py:45).

    ordering="multicolor" (solve mode): IC(0) of the matrix with its unknowns listed colour by colour (red-black for
    every 5- / 7-point grid) -- a DIFFERENT preconditioner from the reference's (another elimination order, ~+20 %
    iterations) whose two triangular solves are a handful of wide parallel sweeps instead of hundreds of dependent
    levels; `CsrSystem.precond_ordering()` returns the ordering.  See dpcg_set_precond_ic0_ordered in include/dpcg.h.
    """

    def __init__(self, mode: str = "solve", ordering: str = "caller"):
        if mode not in ("solve", "multiply"):
            raise ValueError("mode must be 'solve' or 'multiply'")
        if ordering not in ("caller", "multicolor"):
            raise ValueError("ordering must be 'caller' or 'multicolor'")
        if ordering == "multicolor" and mode != "solve":
            raise ValueError("the multicolour ordering serves the triangular solves: mode='solve'")
        self.mode = L.PRECOND_LLT_SOLVE if mode == "solve" else L.PRECOND_LLT_MULTIPLY
        self.ordering = L.ORDER_MULTICOLOR if ordering == "multicolor" else L.ORDER_CALLER

    def _attach(self, system):
        L.check(L.lib().dpcg_set_precond_ic0_ordered(system._h, self.mode, self.ordering, _stream()))

    def __matmul__(self, r):
        raise TypeError("IC0 needs the system matrix: attach it with CsrSystem.set_preconditioner")


class ICholT(Preconditioner):
    """`ilupp.icholt(A, add_fill_in, threshold)` -- what the reference's harness runs by default, with `add_fill_in=1,
    threshold=0.1` (test.py:81-88) -- factored on the device the way ILU++ defines it (Saad's dual-threshold rule on the lower
    triangle): per column the candidates below threshold * ||column||_2 are dropped and of the rest the
    nnz(A[k+1:, k]) + add_fill_in largest are kept.  The ilupp binary is not available: the factor equals the restatement of
    the published algorithm (oracle/oracle.py::icholt) bit for bit, PARITY UNPINNED against ilupp's own output.
    mode="multiply" is the reference's use (it multiplies by L L^T, test.py:88), mode="solve" applies the factor by triangular
    solves.  See dpcg_set_precond_icholt in include/dpcg.h for the limits (64 kept entries per row / column)."""

    def __init__(self, mode: str = "multiply", add_fill_in: int = 1, threshold: float = 0.1):
        if mode not in ("solve", "multiply"):
            raise ValueError("mode must be 'solve' or 'multiply'")
        if add_fill_in < 0 or not threshold >= 0:
            raise ValueError("add_fill_in >= 0 and threshold >= 0")
        self.mode = L.PRECOND_LLT_SOLVE if mode == "solve" else L.PRECOND_LLT_MULTIPLY
        self.add_fill_in, self.threshold = int(add_fill_in), float(threshold)

    def _attach(self, system):
        L.check(L.lib().dpcg_set_precond_icholt(system._h, self.mode, self.add_fill_in, self.threshold, _stream()))

    def __matmul__(self, r):
        raise TypeError("ICholT needs the system matrix: attach it with CsrSystem.set_preconditioner")


class ICT(Preconditioner):
    """Thresholded incomplete Cholesky on a STATIC pattern -- tril(A) plus level-1 fill -- with MATLAB's 'ict' drop rule
    (contract: oracle/oracle.py::ict).  Not what `ilupp.icholt` computes (that is `ICholT`: a per-column entry count instead
    of a fill level); kept because a pattern known before the values lets the factorisation run level-parallel at any size.
    mode="multiply" multiplies by L L^T, mode="solve" applies the factor by triangular solves."""

    def __init__(self, mode: str = "multiply", fill_in: int = 1, threshold: float = 0.1):
        if mode not in ("solve", "multiply"):
            raise ValueError("mode must be 'solve' or 'multiply'")
        if fill_in < 0 or not threshold >= 0:
            raise ValueError("fill_in >= 0 and threshold >= 0")
        self.mode = L.PRECOND_LLT_SOLVE if mode == "solve" else L.PRECOND_LLT_MULTIPLY
        self.fill_in, self.threshold = int(fill_in), float(threshold)

    def _attach(self, system):
        L.check(L.lib().dpcg_set_precond_ict(system._h, self.mode, self.fill_in, self.threshold, _stream()))

    def __matmul__(self, r):
        raise TypeError("ICT needs the system matrix: attach it with CsrSystem.set_preconditioner")


class _DevArray:
    """Zero-copy view of a device buffer the library hands to a callback (CUDA array interface, fp64 vector)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class OperatorPreconditioner(Preconditioner):
    """Any object with `__matmul__` as M -- all the reference's loop asks for (`zk = M @ rk`, cg.py:61,81).

    SLOW PATH, by construction: the library calls back into Python once per update (`dpcg_set_precond_callback`),
    `op @ r` runs as whatever the object does (it receives a CUDA fp64 tensor in the caller's numbering and may
    return a CUDA or a CPU tensor / array), and the updates cannot be replayed as a hipGraph.  SpMV, dot products and
    vector updates remain the HIP kernels.  Prefer a matrix, a factor or one of the `Preconditioner` classes.
    """

    def __init__(self, op):
        if not hasattr(op, "__matmul__"):
            raise TypeError("an operator preconditioner needs __matmul__")
        self.op = op
        self.error: BaseException | None = None
        self.calls = 0

        def _apply(_user, r_ptr, z_ptr, n, _stream_ptr):
            try:
                if self.error is not None:
                    return
                dev = torch.device("cuda", torch.cuda.current_device())
                r = torch.as_tensor(_DevArray(r_ptr, n), device=dev)
                z = torch.as_tensor(_DevArray(z_ptr, n), device=dev)
                out = self.op @ r
                if isinstance(out, np.ndarray):
                    out = torch.from_numpy(out)
                z.copy_(torch.as_tensor(out).reshape(-1).to(device=dev, dtype=torch.float64))
                self.calls += 1
            except BaseException as exc:  # noqa: BLE001 - must not propagate through the C frame; re-raised after the solve
                self.error = exc

        self._fn = L.PRECOND_FN(_apply)      # kept alive with the object

    def _attach(self, system):
        self.error = None
        L.check(L.lib().dpcg_set_precond_callback(system._h, self._fn, None))

    def __matmul__(self, r):
        return self.op @ r


def as_preconditioner(M, n: int) -> Preconditioner:
    """Map whatever the reference's callers pass as `M` to an apply mode.

    None -> identity; a Preconditioner -> itself; a matrix (torch sparse/dense, scipy, numpy) ->
    Jacobi when it is exactly diagonal (bit-identical to the CSR product, one term per row),
    otherwise an explicit CSR multiply; any other object with `__matmul__` -> `OperatorPreconditioner`
    (the reference's operator protocol, honoured through a per-update callback: slow path).
    """
    if M is None:
        return Identity()
    if isinstance(M, Preconditioner):
        return M
    if isinstance(M, CsrSystem):
        raise TypeError("pass the preconditioner as a matrix or a Preconditioner, not a CsrSystem")
    try:
        space, rp, ci, v, m = csr_arrays(M)
    except TypeError:
        if hasattr(M, "__matmul__"):
            return OperatorPreconditioner(M)
        raise
    if m != n:
        raise ValueError("preconditioner size mismatch")
    if diagonal_only(rp, ci, n):
        return Jacobi(v)
    pc = CsrPreconditioner.__new__(CsrPreconditioner)
    pc.space, pc.rowptr, pc.col, pc.val, pc.n = space, rp, ci, v, m
    return pc


# --------------------------------------------------------------------------------------------
# the system operator
# --------------------------------------------------------------------------------------------
@dataclass
class SolveResult:
    x: torch.Tensor
    iterations: int
    status: int          # 0 converged, 1 max_iter, 2 breakdown
    final_res: float     # last tested <r,r>/<b,b>
    seconds: float       # wall time of the iteration loop, device-synchronised (cg.py:69,88)
    res_history: np.ndarray
    err_history: np.ndarray | None = None


_REORDER_MODES = {None: L.REORDER_NONE, False: L.REORDER_NONE, "none": L.REORDER_NONE, "auto": L.REORDER_AUTO,
                  "rcm": L.REORDER_ALWAYS, True: L.REORDER_ALWAYS, "regions": L.REORDER_REGIONS}


class CsrSystem:
    """The operator A of `A @ v` (cg.py:60,75) as a CSR matrix resident in HBM.

    rowptr/col int32, val float64 or float32 CUDA tensors are borrowed (kept alive here).

    `reorder`: "auto" (default) lets the library iterate on P A P^T when the numbering scatters neighbours (large system,
    x-tile plan failed): in reverse Cuthill-McKee order when the measured x-gather traffic is > 4x the bytes used, in the
    cheap region-by-region order when only some row blocks are too scattered (OpenFOAM's numbering after refinement: BASELINE
    config 3) and the x-tile plan takes the result; "rcm" / "regions" force one, None / "none" never reorders.  Everything the caller
    passes or receives -- b, x0, x, `@`, dinv, M, L -- stays in the caller's numbering (`dpcg_reorder`, include/dpcg.h).
    """

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, n: int, reorder="auto"):
        if not (rowptr.is_cuda and col.is_cuda and val.is_cuda):
            raise ValueError("CsrSystem expects CUDA tensors; use CsrSystem.from_host / from_any for host data")
        if rowptr.dtype != torch.int32 or col.dtype != torch.int32:
            raise ValueError("rowptr/col must be int32")
        if val.dtype not in (torch.float64, torch.float32):
            raise ValueError("val must be float64 or float32")
        if rowptr.numel() != n + 1 or col.numel() != val.numel():
            raise ValueError("inconsistent CSR sizes")
        self._keep = (rowptr.contiguous(), col.contiguous(), val.contiguous())
        self.device = val.device
        self.n = int(n)
        self.nnz = int(col.numel())
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib().dpcg_create(C.byref(self._h), self.n, self.nnz, _dev_ptr(self._keep[0]),
                                        _dev_ptr(self._keep[1]), _dev_ptr(self._keep[2]),
                                        L.F64 if val.dtype == torch.float64 else L.F32, L.DEVICE, 0, _stream()))
        self._precond: Preconditioner | None = None
        self._reorder(reorder)

    def _reorder(self, mode) -> None:
        if mode not in _REORDER_MODES:
            raise ValueError("reorder must be 'auto', 'rcm', 'regions' or None")
        applied = C.c_int(0)
        if _REORDER_MODES[mode] != L.REORDER_NONE:
            with torch.cuda.device(self.device):
                L.check(L.lib().dpcg_reorder(self._h, _REORDER_MODES[mode], _stream(), C.byref(applied)))
        self.reordered = bool(applied.value)

    def update_values(self, values) -> None:
        """New matrix values on the SAME sparsity pattern (the next pressure system of one mesh), in the order of the arrays
        the system was created from.  The SpMV plan and the reordering are kept; the preconditioner is dropped -- attach one
        again.  (No reference counterpart: the reference builds a tensor per sample, test.py:61-68.)"""
        if isinstance(values, torch.Tensor) and values.is_cuda:
            v = values.detach().contiguous()
            if v.dtype not in (torch.float32, torch.float64):
                v = v.to(torch.float64)
            if v.numel() != self.nnz:
                raise ValueError(f"expected {self.nnz} values")
            if self._keep and (v.dtype != torch.float64 or v.data_ptr() % 16):
                v = v.to(torch.float64).clone()
            with torch.cuda.device(self.device):
                L.check(L.lib().dpcg_update_values(self._h, _dev_ptr(v), L.F64 if v.dtype == torch.float64 else L.F32,
                                                   L.DEVICE, _stream()))
            if self._keep:                      # a system created from device arrays borrows them: borrow the new ones likewise --
                self._keep = (self._keep[0], self._keep[1], v)      # only once the call has succeeded (the handle still points at the old buffer otherwise)
        else:
            a = values.detach().cpu().numpy() if isinstance(values, torch.Tensor) else np.asarray(values)
            dt = L.F32 if a.dtype == np.float32 else L.F64
            a = np.ascontiguousarray(a, dtype=np.float32 if dt == L.F32 else np.float64)
            if a.size != self.nnz:
                raise ValueError(f"expected {self.nnz} values")
            if self._keep:                      # borrowed device arrays: the new values must live on the device too
                return self.update_values(torch.from_numpy(a.astype(np.float64)).to(self.device))
            with torch.cuda.device(self.device):
                L.check(L.lib().dpcg_update_values(self._h, _np_ptr(a), dt, L.HOST, _stream()))
        self._precond = None

    def permutation(self):
        """perm (numpy int32, perm[new] = old) of a reordered system -- row `new` of the matrix the library iterates on is
        the caller's row `old` -- or None."""
        flag = C.c_int(0)
        perm = np.empty(self.n, dtype=np.int32)
        L.check(L.lib().dpcg_get_permutation(self._h, C.byref(flag), _np_ptr(perm), None))
        return perm if flag.value else None

    # -- constructors ----------------------------------------------------------------------
    @classmethod
    def from_host(cls, rowptr: np.ndarray, col: np.ndarray, val: np.ndarray, n: int, device=None, reorder="auto") -> "CsrSystem":
        self = cls.__new__(cls)
        self._keep = ()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n, self.nnz = int(n), int(len(col))
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        dt = L.F32 if val.dtype == np.float32 else L.F64
        val = np.ascontiguousarray(val, dtype=np.float32 if dt == L.F32 else np.float64)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib().dpcg_create(C.byref(self._h), self.n, self.nnz, _np_ptr(rowptr), _np_ptr(col), _np_ptr(val),
                                        dt, L.HOST, 1, _stream()))
        self._precond = None
        self._reorder(reorder)
        return self

    @classmethod
    def from_any(cls, A, device=None, reorder="auto"):
        if isinstance(A, CsrSystem):
            return A
        space, rp, ci, v, n = csr_arrays(A)
        if space == "host":
            return cls.from_host(rp, ci, v, n, device, reorder)
        return cls(rp, ci, v, n, reorder)

    # -- bookkeeping -------------------------------------------------------------------------
    @property
    def shape(self):
        return (self.n, self.n)

    def info(self) -> dict:
        n, nnz, pn = C.c_int64(), C.c_int64(), C.c_int64()
        k, pk, ll, lu = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.check(L.lib().dpcg_get_info(self._h, C.byref(n), C.byref(nnz), C.byref(k), C.byref(pk), C.byref(pn),
                                      C.byref(ll), C.byref(lu)))
        flag, ratio = C.c_int(0), C.c_double(0.0)
        L.check(L.lib().dpcg_get_permutation(self._h, C.byref(flag), None, C.byref(ratio)))
        return {"n": n.value, "nnz": nnz.value, "spmv_kernel": ("stream", "vector", "tile")[k.value & 15],
                "two_kernel_updates": bool(k.value & 16), "spmv_nt": bool(k.value & 32), "spmv_mixed_tiles": bool(k.value & 64),
                "spmv_cyclic": bool(k.value & 128), "reordered": bool(flag.value),
                "gather_ratio": ratio.value,
                "precond": pk.value, "precond_nnz": pn.value, "levels_lower": ll.value, "levels_upper": lu.value}

    def reduction_geometry(self) -> dict:
        """How the handle's kernels sum their dot products (dpcg_get_reduction_geometry): what a checker needs to add in the same
        order (the tests' CPU restatement takes it as `device_tree=`)."""
        out = (C.c_int32 * 16)()
        L.check(L.lib().dpcg_get_reduction_geometry(self._h, out))
        return {"spmv_grid": out[0], "nrb": out[1], "cyclic": out[2], "vec_grid": out[3], "two_kernel_updates": bool(out[4]),
                "spmv_kernel": ("stream", "vector", "tile")[out[5] & 255], "spmv_tpr": out[5] >> 8, "small_threads": out[6],
                "team_eligible": bool(out[7]), "team_by_default": out[7] == 2, "rz_kind": out[8], "m_grid": out[9], "m_nrb": out[10], "m_cyclic": out[11] & 255,
                "m_tpr": (out[11] >> 8) & 255, "mt_tpr": (out[11] >> 16) & 255, "sweep_grid": out[13],
                "sweep_modes": [(out[14] >> (2 * k)) & 3 for k in range(out[12])], "sweep_paired": bool(out[15])}

    def chip_info(self) -> dict:
        """The whole-chip solve of cache-sized systems (dpcg_get_chip_info): eligibility, the geometry a checker needs to add the dot
        products in the kernel's order, and -- with DPCG_CHIP_TRACE=1 -- microseconds per update by phase of the last such solve."""
        out = (C.c_int32 * 8)()
        tr = (C.c_double * 8)()
        L.check(L.lib().dpcg_get_chip_info(self._h, out, tr))
        return {"chip_eligible": bool(out[0]), "chip_by_default": out[0] == 2, "workgroups": out[1], "threads": out[2],
                "rows_per_workgroup": out[3], "max_row_len": out[4], "max_band": out[5], "groups_on_one_xcd": bool(out[6] & 1),
                "lanes_per_row": (out[6] >> 8) & 255,
                "kernel_ms": out[7] * 1e-6,
                "trace_us": {"spmv": tr[0], "sum_pq": tr[1], "update_publish": tr[2], "sum_rz_rr": tr[3], "loop": tr[4],
                             "wait_pq": tr[5], "wait_rz": tr[6], "updates": int(tr[7])}}

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            L.lib().dpcg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _vec(self, v) -> torch.Tensor:
        t = _to_device_f64(v, self.device)
        if t.dim() != 1 or t.numel() != self.n:
            raise ValueError(f"expected a vector of length {self.n}")
        return t

    # -- operators -------------------------------------------------------------------------
    def __matmul__(self, v) -> torch.Tensor:
        """y = A v (cg.py:60,75)."""
        x = self._vec(v)
        y = torch.empty_like(x)
        with torch.cuda.device(self.device):
            L.check(L.lib().dpcg_spmv(self._h, _dev_ptr(x), _dev_ptr(y), _stream()))
        return y

    def spmv_f32(self, v: torch.Tensor) -> torch.Tensor:
        x = v.to(device=self.device, dtype=torch.float32).contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(self.device):
            L.check(L.lib().dpcg_spmv_f32(self._h, _dev_ptr(x), _dev_ptr(y), _stream()))
        return y

    def set_preconditioner(self, M) -> Preconditioner:
        pc = as_preconditioner(M, self.n)
        with torch.cuda.device(self.device):
            pc._attach(self)
        self._precond = pc
        return pc

    def precond_apply(self, r) -> torch.Tensor:
        """z = M r (cg.py:61,81) for the attached preconditioner."""
        rv = self._vec(r)
        z = torch.empty_like(rv)
        with torch.cuda.device(self.device):
            L.check(L.lib().dpcg_precond_apply(self._h, _dev_ptr(rv), _dev_ptr(z), _stream()))
        return z

    def sptrsv(self, rhs, upper: bool) -> torch.Tensor:
        rv = self._vec(rhs)
        out = torch.empty_like(rv)
        with torch.cuda.device(self.device):
            L.check(L.lib().dpcg_sptrsv(self._h, 1 if upper else 0, _dev_ptr(rv), _dev_ptr(out), _stream()))
        return out

    def precond_ordering(self):
        """(n_colors, perm) of the attached factor's own numbering: perm[k] = the caller's row at factor position k
        (identity and 0 colours unless the factor was built with ordering="multicolor")."""
        nc = C.c_int(0)
        perm = np.empty(self.n, dtype=np.int32)
        L.check(L.lib().dpcg_get_precond_ordering(self._h, C.byref(nc), _np_ptr(perm)))
        return nc.value, perm

    def factor(self):
        """The attached L factor as host CSR arrays (rowptr, col, val), in the factor's numbering (`precond_ordering`)."""
        nnz = self.info()["precond_nnz"]
        rp = np.empty(self.n + 1, dtype=np.int32)
        ci = np.empty(nnz, dtype=np.int32)
        v = np.empty(nnz, dtype=np.float64)
        L.check(L.lib().dpcg_get_factor(self._h, _np_ptr(rp), _np_ptr(ci), _np_ptr(v)))
        return rp, ci, v

    def spmv_dot_bench(self, repeats: int = 100) -> float:
        """Average milliseconds of the in-PCG SpMV+<p,Ap> kernel over `repeats` launches (HIP events)."""
        x = torch.rand(self.n, device=self.device, dtype=torch.float64)
        y = torch.empty_like(x)
        ms = C.c_float()
        with torch.cuda.device(self.device):
            L.check(L.lib().dpcg_spmv_dot_bench(self._h, _dev_ptr(x), _dev_ptr(y), repeats, C.byref(ms), _stream()))
        return float(ms.value)

    def solve(self, b, x0=None, *, rtol_sq: float = 1e-8, atol_sq: float = 0.0, max_iter: int = 1024, flags: int = 0,
              x_true=None, want_history: bool = True) -> SolveResult:
        """Run the PCG loop of cg.py:58-90 on the GPU (see dpcg_solve in include/dpcg.h)."""
        bv = self._vec(b)
        x0v = None if x0 is None else self._vec(x0)
        xt = None if x_true is None else self._vec(x_true)
        x = torch.empty_like(bv)
        hist = np.full(max_iter + 1, np.nan) if want_history else None
        err = np.full(max_iter + 1, np.nan) if xt is not None else None
        iters, res, sec = C.c_int(), C.c_double(), C.c_double()
        with torch.cuda.device(self.device):
            status = L.check(L.lib().dpcg_solve(
                self._h, _dev_ptr(bv), _dev_ptr(x0v), _dev_ptr(x), rtol_sq, atol_sq, int(max_iter), int(flags),
                _stream(), C.byref(iters), C.byref(res), C.byref(sec), _np_ptr(hist), _dev_ptr(xt), _np_ptr(err)))
        if isinstance(self._precond, OperatorPreconditioner) and self._precond.error is not None:
            err, self._precond.error = self._precond.error, None
            raise err
        k = iters.value
        return SolveResult(x, k, status, res.value, sec.value, hist[: k + 1] if hist is not None else np.empty(0),
                           err[: k + 1] if err is not None else None)


def dot(a: torch.Tensor, b: torch.Tensor) -> float:
    """<a,b> by the deterministic two-stage HIP reduction (torch.inner at cg.py:17,76,78,82)."""
    a = a.to(torch.float64).contiguous()
    b = b.to(device=a.device, dtype=torch.float64).contiguous()
    out = C.c_double()
    with torch.cuda.device(a.device):
        L.check(L.lib().dpcg_dot(a.numel(), _dev_ptr(a), _dev_ptr(b), C.byref(out), _stream()))
    return float(out.value)


def stream_bench(n_read: int = 2, write: bool = True, out_bytes: int = 1 << 27, repeats: int = 10, nontemporal: bool = False,
                 walk: bool = False) -> float:
    """GB/s (reads + writes) of the library's own streaming kernel on this box: per 16 bytes written, `n_read` x 16
    contiguous bytes are read (or only reduced when write=False) -- the measured HBM ceiling next to the 8 TB/s spec
    (SURVEY.md 8-d2).  n_read = 11 with write is the read:write ratio of a 7-point CSR SpMV."""
    ms, moved = C.c_float(), C.c_int64()
    # walk: the streams walked together by the whole grid instead of one slab per workgroup (a copy then is the guide's
    # float4-copy shape: one 16-byte element per thread)
    L.check(L.lib().dpcg_stream_bench(int(n_read), 1 if write else 0, (1 if nontemporal else 0) | (2 if walk else 0), int(out_bytes), int(repeats),
                                      C.byref(ms), C.byref(moved), _stream()))
    return moved.value / (ms.value * 1e-3) / 1e9


def release_cached_memory() -> None:
    """Return the device blocks the library keeps between setups (see include/dpcg.h) to the driver."""
    L.check(L.lib().dpcg_release_cached_memory())
