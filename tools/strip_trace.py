"""DPCG_STRIP_TRACE=1 python tools/strip_trace.py dim size [repeats]: the per-strip timeline of ONE lower and ONE upper strip-pipelined solve
(IC(0) factor of a natural-order grid) on stderr."""
import sys
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

dim, size = int(sys.argv[1]), int(sys.argv[2])
s = poisson.poisson_system(dim, size)
s.set_preconditioner(D.IC0("solve"))
r = poisson.rhs(s.n, 0)
print("==== traced", file=sys.stderr, flush=True)
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):     # repeated: the later solves find their records in L2 / MALL
    s.sptrsv(r, False)
    torch.cuda.synchronize()
