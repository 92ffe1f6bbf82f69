"""Mid-size systems: the one-launch team solve (dpcg_team.hip) against the multi-launch path -- BASELINE config 2's 256^2
Jacobi solve alone, and eight such systems as one batch (one team per XCD).   python tools/team_probe.py"""
import time

import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from deeppreconditioning_amd.batch import solve_batch

for dim, n in ((2, 100), (2, 256), (3, 40)):
    s = poisson.poisson_system(dim, n)
    s.set_preconditioner(D.Jacobi())
    b = poisson.rhs(s.n, 0)
    for label, flags in (("team", D._lib.TEAM), ("multi-launch", D._lib.NO_TEAM)):
        s.solve(b, want_history=False, flags=flags)
        best = None
        for _ in range(5):
            r = s.solve(b, want_history=False, flags=flags)
            best = r if best is None or r.seconds < best.seconds else best
        print(f"poisson{dim}d_{n} rows {s.n:6d} {label:13s}: {best.iterations} its, {best.seconds * 1e3:7.3f} ms = "
              f"{best.seconds / best.iterations * 1e6:6.2f} us/update = {best.iterations / best.seconds / 1e3:7.1f} K it/s", flush=True)
    group = [poisson.poisson_system(dim, n) for _ in range(8)]
    for g in group:
        g.set_preconditioner(D.Jacobi())
    rhs = [poisson.rhs(g.n, i) for i, g in enumerate(group)]
    for label, flags in (("team", 0), ("multi-launch", D._lib.NO_TEAM)):
        solve_batch(group, rhs, flags=flags)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = solve_batch(group, rhs, flags=flags)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        its = sum(r.iterations for r in res)
        print(f"8 x poisson{dim}d_{n} batch {label:13s}: {its} its in {dt * 1e3:7.3f} ms = {its / dt / 1e3:8.1f} K it/s aggregate", flush=True)
    for g in group:
        g.close()
    s.close()
