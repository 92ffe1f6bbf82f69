import os, sys, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
for name, mk in (("poisson3d_100", lambda: poisson.poisson_system(3, 100)), ("poisson2d_1024", lambda: poisson.poisson_system(2, 1024))):
    s = mk(); s.set_preconditioner(D.Jacobi())
    b = poisson.rhs(s.n, 0)
    ms = min(s.spmv_dot_bench(200) for _ in range(3))
    s.solve(b, want_history=False)
    r = min((s.solve(b, want_history=False) for _ in range(5)), key=lambda r: r.seconds)
    print(os.environ.get("DPCG_SPMV_WG_PER_CU", "8"), name, f"K1 {ms*1e3:.2f} us  PCG {r.iterations / r.seconds:.0f} it/s", flush=True)
    s.close()
