"""Config 2 (256^2) Jacobi PCG, repeated: seconds per solve and per update, default flags and the launch-form flags."""
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

s = poisson.poisson_system(2, 256)
b = poisson.rhs(s.n, 0)
s.set_preconditioner(D.Jacobi())
for name, flags in (("default", 0), ("no_graph", D._lib.NO_GRAPH), ("no_fuse", D._lib.NO_FUSE | D._lib.NO_SMALL)):
    for rep in range(4):
        r = s.solve(b, want_history=False, flags=flags)
        print(f"{name:9s} rep {rep}: {r.iterations} its {r.seconds * 1e3:8.3f} ms = {r.seconds / r.iterations * 1e6:6.2f} us/update", flush=True)
