"""Name-compatible shim: put `compat/` on PYTHONPATH and the reference's import lines resolve to the MI355X path.

    from uibk.deep_preconditioning.cg import preconditioned_conjugate_gradient     # test.py:20, train.py:16
    from uibk.deep_preconditioning.utils import sparse_matvec_mul                  # metrics.py:7
    import uibk.deep_preconditioning.model as models                               # test.py:19, train.py:15

Only the modules on the solve path are provided (cg, utils, model, metrics); everything re-exports
`deeppreconditioning_amd`.
"""
