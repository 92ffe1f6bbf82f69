"""`uibk.deep_preconditioning.metrics` on the MI355X path: the two losses of the solve / training path (metrics.py:13-55).
`hutchinson_trace` and `condition_loss` (metrics.py:58-100) are outside this path and are not provided."""
from deeppreconditioning_amd.metrics import frobenius_loss, inverse_loss  # noqa: F401
