"""`uibk.deep_preconditioning.cg` on the MI355X path (same names and signatures as the reference's cg.py:15-90)."""
from deeppreconditioning_amd.cg import (conjugate_gradient, preconditioned_conjugate_gradient,  # noqa: F401
                                        stopping_criterion)
