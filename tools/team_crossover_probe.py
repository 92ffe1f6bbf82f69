"""One mid-size system: the one-launch team solve against the multi-launch path, by size (where does a single system switch?).
python tools/team_crossover_probe.py"""
import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

for dim, n in ((2, 80), (2, 100), (2, 128), (2, 160), (2, 200), (2, 230), (2, 256), (3, 20), (3, 25), (3, 30), (3, 35), (3, 40)):
    s = poisson.poisson_system(dim, n)
    s.set_preconditioner(D.Jacobi())
    b = poisson.rhs(s.n, 0)
    out = {}
    for label, flags in (("team", D._lib.TEAM), ("multi", D._lib.NO_TEAM), ("default", 0)):
        s.solve(b, want_history=False, flags=flags)
        best = min((s.solve(b, want_history=False, flags=flags) for _ in range(7)), key=lambda r: r.seconds)
        out[label] = best.seconds / best.iterations * 1e6
    print(f"poisson{dim}d_{n} rows {s.n:6d}: team {out['team']:6.2f} us/update, multi-launch {out['multi']:6.2f}, default {out['default']:6.2f}", flush=True)
    s.close()
