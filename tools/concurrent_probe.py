"""Aggregate PCG iterations/s when several independent systems are solved at once on one GPU (solve_batch, one stream
per system in flight) vs one after another."""
import time
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import batch as B
from deeppreconditioning_amd import poisson

for dim, n, count in [(3, 100, 2), (3, 100, 4), (3, 64, 4), (2, 256, 8)]:
    systems = [poisson.poisson_system(dim, n, device="cuda:0") for _ in range(count)]
    for s in systems:
        s.set_preconditioner(D.Jacobi())
    rhs = [poisson.rhs(s.n, i) for i, s in enumerate(systems)]
    for s, b in zip(systems, rhs):
        s.solve(b, want_history=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = sum(s.solve(b, want_history=False, flags=D._lib.NO_SMALL).iterations for s, b in zip(systems, rhs))
    torch.cuda.synchronize()
    seq = its / (time.perf_counter() - t0)
    for streams in (2, 4):
        B.solve_batch(systems, rhs, n_streams=streams, flags=D._lib.NO_SMALL)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = B.solve_batch(systems, rhs, n_streams=streams, flags=D._lib.NO_SMALL)
        torch.cuda.synchronize()
        conc = sum(r.iterations for r in out) / (time.perf_counter() - t0)
        print(f"poisson{dim}d_{n} x{count}: sequential {seq:9.0f} it/s   {streams} streams {conc:9.0f} it/s   ratio {conc / seq:.2f}")
    for s in systems:
        s.close()
