"""`uibk.deep_preconditioning.model` on PyTorch-ROCm without spconv: `PreconditionerNet` (model.py:13-59), the network of the solve
path.  `PreconditionerSparseUNet` (model.py:62-179) is outside this path and is not provided."""
from deeppreconditioning_amd.model import PreconditionerNet, SparseConv2d, SparseSequential  # noqa: F401
