"""`uibk.deep_preconditioning.metrics` (metrics.py:13-55) on the MI355X path."""
from deeppreconditioning_amd.metrics import frobenius_loss, inverse_loss  # noqa: F401
