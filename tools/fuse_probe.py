"""Two-kernel vs three-kernel updates: Jacobi-PCG iterations/s over system sizes (decides the row threshold)."""
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

L = D._lib
for dim, n in [(2, 80), (2, 128), (2, 256), (2, 384), (2, 512), (3, 64), (2, 768), (3, 80), (2, 1024)]:
    s = poisson.poisson_system(dim, n, device="cuda:0")
    s.set_preconditioner(D.Jacobi())
    b = poisson.rhs(s.n, 0)
    out = []
    for flags in (L.NO_SMALL, L.NO_SMALL | L.NO_FUSE):
        s.solve(b, want_history=False, flags=flags, max_iter=400)
        best = 0.0
        for _ in range(3):
            r = s.solve(b, want_history=False, flags=flags, max_iter=400)
            best = max(best, r.iterations / r.seconds)
        out.append(best)
    print(f"poisson{dim}d_{n:5d} rows {s.n:8d} kernel {s.info()['spmv_kernel']:7s} two-kernel {out[0]:10.0f} it/s   three-kernel {out[1]:10.0f} it/s   ratio {out[0] / out[1]:.3f}")
