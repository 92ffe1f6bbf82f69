"""MI355X-native preconditioned-CG solve path with the API of uibk.deep_preconditioning.

    from deeppreconditioning_amd.cg import preconditioned_conjugate_gradient

The solver kernels live in `csrc/` (HIP, gfx950) behind the C ABI of `include/dpcg.h`.
"""

from . import _lib  # noqa: F401
from .operators import (IC0, ICT, ICholT, CsrPreconditioner, CsrSystem, Identity, Jacobi, LLtMultiply, LLtSolve,  # noqa: F401
                        OperatorPreconditioner, Preconditioner, SolveResult, as_preconditioner, csr_arrays, dot)

__all__ = ["CsrSystem", "Preconditioner", "Identity", "Jacobi", "CsrPreconditioner", "LLtMultiply", "LLtSolve", "IC0", "ICT", "ICholT", "OperatorPreconditioner",
           "SolveResult", "as_preconditioner", "csr_arrays", "dot"]
