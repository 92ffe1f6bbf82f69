import sys, os
sys.path.insert(0, os.getcwd())
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
s = poisson.poisson_system(3, int(sys.argv[1]) if len(sys.argv) > 1 else 100); s.set_preconditioner(D.Jacobi()); b = poisson.rhs(s.n, 0)
for _ in range(3):
    r = s.solve(b, want_history=False)
print(r.iterations, r.seconds / r.iterations * 1e6)
print(s.chip_info()["trace_us"])
