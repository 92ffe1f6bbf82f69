"""`uibk.deep_preconditioning.metrics` on the MI355X path: the two losses of the solve / training path (metrics.py:13-55).
`hutchinson_trace` and `condition_loss` (metrics.py:58-100; scripts/compare_meshes.py:65 calls the latter) are outside this path
(SURVEY.md 2 #5): plain torch restatements, resolved lazily."""
from deeppreconditioning_amd.metrics import frobenius_loss, inverse_loss  # noqa: F401


def __getattr__(name):
    import deeppreconditioning_amd.metrics as _m
    return getattr(_m, name)
