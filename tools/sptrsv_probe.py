"""Times z = (L L^T)^{-1} r (two level-scheduled SpTRSVs) and the IC(0)-solve PCG on Poisson systems."""
import sys
import time
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

for dim, n in [(2, 256), (2, 1024), (3, 64), (3, 100)]:
    s = poisson.poisson_system(dim, n, device="cuda:0")
    s.set_preconditioner(D.IC0(mode="solve"))
    info = s.info()
    r = poisson.rhs(s.n, 0)
    z = s.precond_apply(r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        z = s.precond_apply(r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    res = s.solve(r, want_history=False)
    res = s.solve(r, want_history=False)
    print(f"poisson{dim}d_{n}: levels {info['levels_lower']}  apply {dt * 1e3:8.3f} ms  "
          f"({dt * 1e6 / (2 * info['levels_lower']):.3f} us/level)  PCG {res.iterations} its {res.seconds * 1e3:8.2f} ms "
          f"checksum {float(z.double().sum()):.15e}")
