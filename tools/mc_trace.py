"""Workload for `rocprofv3 --kernel-trace --stats`: preconditioner applies of IC(0) in multicolour order (2 levels) and in
the caller's order on the 1M-DoF systems.   python tools/mc_trace.py [natural|scrambled]"""
import sys
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

which = sys.argv[1] if len(sys.argv) > 1 else "natural"
s = poisson.poisson_system(3, 100) if which == "natural" else D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 100, 0))
r = poisson.rhs(s.n, 0)
for pc in (D.IC0("solve", ordering="multicolor"), D.IC0("solve")):
    s.set_preconditioner(pc)
    poisson.poisson_csr(2, 8)                      # segment marker in the trace (k_gen_poisson)
    for _ in range(10):
        s.precond_apply(r)
    torch.cuda.synchronize()
