"""Workload for the rocprofv3 passes on the whole-chip solve (dpcg_chip.hip): full solves of the headline system (poisson3d_100, Jacobi,
187 updates) and of poisson2d_1024 (capped at 300 updates), through the plain call; prints `name updates` per system."""
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from deeppreconditioning_amd import poisson  # noqa: E402

torch.cuda.set_device(0)
for dim, n, cap in ((3, 100, 1024), (2, 1024, 300)):
    poisson.poisson_csr(2, 8)                      # segment marker (k_gen_poisson)
    s = poisson.poisson_system(dim, n)
    s.set_preconditioner(D.Jacobi())
    assert s.chip_info()["chip_by_default"]
    b = poisson.rhs(s.n, 0)
    for _ in range(6):
        r = s.solve(b, max_iter=cap, want_history=False)
    print(f"poisson{dim}d_{n} {r.iterations} {s.n} {s.nnz}", flush=True)
    del s
torch.cuda.synchronize()
