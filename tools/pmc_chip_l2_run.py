"""Workload for the L2-level counter passes on the whole-chip solve (tools/pmc_chip_l2.sh): the headline system (poisson3d_100, Jacobi, 187
updates) through the plain call, then -- for the kernel trace -- its two development variants (DPCG_CHIP_BENCH=1: no gathers, =3: gathers
issued out of range), the same number of updates each; prints `name updates n nnz`."""
import os
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from deeppreconditioning_amd import poisson  # noqa: E402

torch.cuda.set_device(0)
s = poisson.poisson_system(3, 100)
s.set_preconditioner(D.Jacobi())
assert s.chip_info()["chip_by_default"]
b = poisson.rhs(s.n, 0)
for _ in range(6):
    r = s.solve(b, want_history=False)
its = r.iterations
if "--variants" in sys.argv:
    for env in ("1", "3"):
        os.environ["DPCG_CHIP_BENCH"] = env
        for _ in range(6):
            s.solve(b, max_iter=its, want_history=False)
    os.environ.pop("DPCG_CHIP_BENCH", None)
print(f"poisson3d_100 {its} {s.n} {s.nnz}", flush=True)
torch.cuda.synchronize()
