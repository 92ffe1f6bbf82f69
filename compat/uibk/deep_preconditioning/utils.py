"""`uibk.deep_preconditioning.utils` on the MI355X path (utils.py:15-76)."""
from deeppreconditioning_amd.utils import SparseBatch, benchmark_cg, sparse_matvec_mul  # noqa: F401
