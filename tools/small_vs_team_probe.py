import torch, deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
for n in (49, 56, 60, 64, 70, 75, 78):
    s = poisson.poisson_system(2, n); s.set_preconditioner(D.Jacobi()); b = poisson.rhs(s.n, 0)
    out = {}
    for label, flags in (("small", D._lib.NO_TEAM), ("team", D._lib.TEAM)):
        s.solve(b, want_history=False, flags=flags)
        best = min((s.solve(b, want_history=False, flags=flags) for _ in range(9)), key=lambda r: r.seconds)
        out[label] = best.seconds / best.iterations * 1e6
    print(f"poisson2d_{n} rows {s.n}: one-workgroup kernel {out['small']:.2f} us/update, team {out['team']:.2f}", flush=True)
    s.close()
