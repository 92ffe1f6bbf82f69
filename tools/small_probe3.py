import sys, time, torch
sys.path.insert(0, "/root/repo")
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from deeppreconditioning_amd.batch import solve_batch
for pcname in ("jacobi", "none"):
    for count in (1, 64, 256):
        systems, rhs = [], []
        for i in range(count):
            S = poisson.poisson_system(2, 49); S.set_preconditioner(D.Jacobi() if pcname == "jacobi" else None); systems.append(S); rhs.append(poisson.rhs(S.n, i))
        solve_batch(systems, rhs, max_iter=100, rtol_sq=0.0)
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = solve_batch(systems, rhs, max_iter=100, rtol_sq=0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{pcname:6s} batch {count:3d}: {dt*1e3:.3f} ms -> {dt/100*1e6:.1f} us per iteration step")
