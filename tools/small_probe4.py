import sys, time, torch
sys.path.insert(0, "/root/repo")
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from deeppreconditioning_amd.batch import solve_batch
for n in (8, 49):
    for count in (1, 16, 64, 256):
        systems, rhs = [], []
        for i in range(count):
            S = poisson.poisson_system(2, n); S.set_preconditioner(D.Jacobi()); systems.append(S); rhs.append(poisson.rhs(S.n, i))
        for its in (100, 400):
            solve_batch(systems, rhs, max_iter=its, rtol_sq=0.0)
            torch.cuda.synchronize(); t0 = time.perf_counter(); out = solve_batch(systems, rhs, max_iter=its, rtol_sq=0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"N={n*n:5d} batch {count:3d} its {its}: {dt*1e3:.3f} ms -> {dt/its*1e6:.1f} us per iteration step")
