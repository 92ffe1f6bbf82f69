"""`uibk.deep_preconditioning.model` on PyTorch-ROCm without spconv (model.py:13-179)."""
from deeppreconditioning_amd.model import (PreconditionerNet, PreconditionerSparseUNet, SparseConv2d,  # noqa: F401
                                           SparseInverseConv2d, SparseSequential, SubMConv2d, sparse_add)
