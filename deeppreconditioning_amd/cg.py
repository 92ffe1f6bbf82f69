"""The conjugate gradient method with optional preconditioners -- GPU drop-in for
`uibk/deep_preconditioning/cg.py` (same names, arguments and return values).

The iteration itself (CSR SpMV, preconditioner apply, fused dot/axpy/residual updates, stopping
test) runs in the hand-written HIP kernels of libdpcg.so; see include/dpcg.h.  Differences from the
reference that a caller can observe:

* `info` (third return value of PCG) is a real status -- 0 converged, 1 `max_iter` reached,
  2 breakdown -- where the reference hard-wires 0 (cg.py:90).
* `A` may be anything `operators.csr_arrays` understands (dense / sparse torch tensors, scipy,
  numpy) or a prepared `CsrSystem`; `M` a matrix, a `Preconditioner`, None, or -- the reference's
  operator protocol, cg.py:61,81 -- any object with `__matmul__`, which is applied through a
  per-update callback (`OperatorPreconditioner`: slow path, the rest of the update stays HIP).
* `x_true` is accepted and ignored by PCG: the reference only uses it for an error history that
  it discards (cg.py:64-67,85-88,90).
"""

from __future__ import annotations

import torch

from . import _lib as L
from .operators import CsrSystem, dot


def stopping_criterion(_, rk, b):
    """<rk,rk>/<b,b>: the SQUARED relative residual (cg.py:15-17), as a 0-d tensor."""
    rk = torch.as_tensor(rk)
    dev = rk.device if rk.is_cuda else torch.device("cuda", torch.cuda.current_device())
    rk = rk.to(dev)
    b = torch.as_tensor(b).to(dev)
    return torch.tensor(dot(rk, rk) / dot(b, b), dtype=torch.float64, device=dev)


def _system_and_device(A, b):
    if isinstance(A, CsrSystem):
        return A
    device = None
    if isinstance(b, torch.Tensor) and b.is_cuda:
        device = b.device
    return CsrSystem.from_any(A, device=device)


def preconditioned_conjugate_gradient(A, b, M, x0=None, x_true=None, rtol=1e-8, max_iter=1024, *,
                                      mixed_precision=False, compact_values=False, details=False):
    """PCG, cg.py:50-90.  Returns `(duration_seconds, iterations, info)`.

    `rtol` is compared with <r,r>/<b,b> (squared ratio, cg.py:71); the first test uses z0
    (cg.py:66).  `duration` covers the iteration loop only, device-synchronised (cg.py:69,88).
    Keyword-only extras: `mixed_precision` runs A@p with fp32 matrix values and an fp32 copy of p
    (fp64 everywhere else, BASELINE config 5); `compact_values` streams the matrix values as fp32
    when that is lossless (the reference's matrices are fp32 data upcast to fp64, test.py:68) --
    bit-identical results, 8 instead of 12 bytes per non-zero; `details=True` returns the full
    `SolveResult` (status 0 converged / 1 max_iter / 2 breakdown, residual history, x).

    The plain return value follows the reference to the letter: `info` is always 0 (cg.py:90), and a NaN
    breakdown (b = 0, NaN input, singular M), on which the reference keeps looping because `nan < rtol` is
    false, reports `max_iter` iterations -- the library stops at the first NaN and says so in `details`.
    """
    del x_true  # unused by the reference's return value
    system = _system_and_device(A, b)
    system.set_preconditioner(M)
    flags = (L.SPMV_F32 if mixed_precision else 0) | (L.VAL32_IF_LOSSLESS if compact_values else 0)
    result = system.solve(b, x0, rtol_sq=float(rtol), max_iter=int(max_iter), flags=flags)
    if details:
        return result
    iterations = int(max_iter) if result.status == L.BREAKDOWN else result.iterations
    return result.seconds, iterations, 0


def conjugate_gradient(A, b, x0=None, x_true=None, rtol=1e-8, max_iter=1024):
    """Unpreconditioned CG, cg.py:20-47.  Returns `(errors, x_hat)` with
    `errors[k] = (A-norm error of x_k or 0, <r_k,r_k>/<b,b>)` as 0-d tensors."""
    system = _system_and_device(A, b)
    system.set_preconditioner(None)
    result = system.solve(b, x0, rtol_sq=float(rtol), max_iter=int(max_iter), flags=L.INIT_CHECK_R, x_true=x_true)
    res = torch.from_numpy(result.res_history.copy()).to(system.device)
    if result.err_history is not None:
        err = torch.from_numpy(result.err_history.copy()).to(system.device)
    else:
        err = torch.zeros_like(res)
    errors = list(zip(err.unbind(0), res.unbind(0)))
    x_hat = result.x
    if isinstance(b, torch.Tensor) and b.dtype != torch.float64:
        x_hat = x_hat.to(b.dtype)  # the reference follows b's dtype (cg.py:22)
    return errors, x_hat
