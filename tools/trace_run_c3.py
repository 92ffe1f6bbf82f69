"""Workload for `rocprofv3 --kernel-trace`: IC(0)-by-triangular-solves PCG on the config-3 stand-in (scrambled 1M-DoF system, library
reordering for the SpMV, factor of the caller's matrix: 19 wide levels) -- where does an update's time go, level by level?"""
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from deeppreconditioning_amd import poisson  # noqa: E402

torch.cuda.set_device(0)
A = poisson.unstructured_like_csr(3, 100, 0)
s = D.CsrSystem.from_any(A)
# `multicolor`: IC(0) of the same system in multicolour order (2 levels: colour-sweep kernels) instead of the caller's order
s.set_preconditioner(D.IC0("solve", ordering="multicolor") if "multicolor" in sys.argv[1:] else D.IC0("solve"))
b = poisson.rhs(s.n, 0)
s.solve(b, max_iter=4, want_history=False)
s.solve(b, max_iter=24, want_history=False, flags=D._lib.NO_GRAPH)
torch.cuda.synchronize()
