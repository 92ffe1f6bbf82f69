"""`uibk.deep_preconditioning.metrics` (metrics.py:13-100) on the MI355X path."""
from deeppreconditioning_amd.metrics import (condition_loss, frobenius_loss, hutchinson_trace,  # noqa: F401
                                              inverse_loss)
