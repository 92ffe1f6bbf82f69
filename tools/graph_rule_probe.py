"""A/B of the graph rule for updates of 25 .. 100 us (DPCG_GRAPH_LONG_US): the 1M-row launch paths -- Jacobi, IC(0) in multicolour order, IC(0) in the
caller's order -- us per update:  DPCG_GRAPH_LONG_US=25 python tools/graph_rule_probe.py"""
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from oracle import oracle as O  # noqa: E402

A = O.unstructured_like(O.poisson3d(100), seed=0)
S = D.CsrSystem.from_any(A)
b = torch.from_numpy(O.rhs(A.shape[0], 0)).cuda()
out = []
for name, pc in (("jacobi", D.Jacobi()), ("ic0 multicolour", D.IC0("solve", ordering="multicolor")), ("ic0 caller", D.IC0("solve"))):
    S.set_preconditioner(pc)
    v = []
    for _ in range(4):
        r = S.solve(b, flags=D._lib.NO_SMALL, want_history=False)
        v.append(r.seconds / r.iterations * 1e6)
    out.append(f"{name}: {np.median(v[1:]):.2f} us per update ({r.iterations} updates)")
print("; ".join(out), flush=True)
