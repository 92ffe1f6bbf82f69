import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from deeppreconditioning_amd.batch import solve_batch
for dim, n in ((2, 8), (2, 16), (2, 32), (2, 49), (2, 64), (2, 78)):
    S = poisson.poisson_system(dim, n); S.set_preconditioner(D.Jacobi()); b = poisson.rhs(S.n, 0)
    S.solve(b, max_iter=200, rtol_sq=0.0)
    t = [S.solve(b, max_iter=200, rtol_sq=0.0, want_history=False) for _ in range(5)]
    r = min(t, key=lambda r: r.seconds)
    t1 = [S.solve(b, max_iter=1, rtol_sq=0.0, want_history=False) for _ in range(5)]
    r1 = min(t1, key=lambda r: r.seconds)
    print(f"N={S.n:5d}: 200 its {r.seconds*1e3:.3f} ms; 1 it {r1.seconds*1e3:.3f} ms -> {(r.seconds-r1.seconds)/199*1e6:.2f} us/it")
for count in (1, 8, 32, 64, 128, 256):
    systems, rhs = [], []
    for i in range(count):
        S = poisson.poisson_system(2, 49); S.set_preconditioner(D.Jacobi()); systems.append(S); rhs.append(poisson.rhs(S.n, i))
    solve_batch(systems, rhs, max_iter=100, rtol_sq=0.0)
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = solve_batch(systems, rhs, max_iter=100, rtol_sq=0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"batch {count:3d} x N=2401, 100 its each: {dt*1e3:.3f} ms -> {dt/100*1e6:.1f} us per iteration step")
