"""`uibk.deep_preconditioning.model` on PyTorch-ROCm without spconv (model.py:13-59)."""
from deeppreconditioning_amd.model import PreconditionerNet, SparseConv2d, SparseSequential  # noqa: F401
