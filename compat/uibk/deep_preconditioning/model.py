"""`uibk.deep_preconditioning.model` on PyTorch-ROCm without spconv: `PreconditionerNet` (model.py:13-59), the network of the solve
path.  `PreconditionerSparseUNet` (model.py:62-179; `getattr(models, params["model"])`, test.py:215, train.py:154) is outside this
path: a plain torch restatement, resolved lazily."""
from deeppreconditioning_amd.model import PreconditionerNet, SparseConv2d, SparseSequential  # noqa: F401


def __getattr__(name):
    import deeppreconditioning_amd.model as _m
    return getattr(_m, name)
