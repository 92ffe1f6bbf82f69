"""How long does the GPU stay slow after the host kept it idle?  Config-2 Jacobi solves (4.4 ms each when warm) right after
t seconds of host-only work."""
import time
import numpy as np, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

s = poisson.poisson_system(2, 256)
b = poisson.rhs(s.n, 0)
s.set_preconditioner(D.Jacobi())
for _ in range(5):
    s.solve(b, want_history=False)
for idle in (0.0, 0.5, 2.0, 8.0):
    t_end = time.perf_counter() + idle
    a = np.random.rand(400, 400)
    while time.perf_counter() < t_end:       # host-only work (a busy host, not a sleeping one)
        a = a @ a
        a /= np.abs(a).max()
    out = []
    for _ in range(6):
        t0 = time.perf_counter()
        r = s.solve(b, want_history=False)
        out.append((r.seconds * 1e3, (time.perf_counter() - t0) * 1e3))
    print(f"after {idle:4.1f} s of host work: solve ms (loop / wall) " + "  ".join(f"{a:6.2f}/{w:6.2f}" for a, w in out), flush=True)
