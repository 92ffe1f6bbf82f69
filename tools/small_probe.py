import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from deeppreconditioning_amd.batch import solve_batch
for dim, n in ((2, 64), (2, 49), (3, 18)):
    S = poisson.poisson_system(dim, n); S.set_preconditioner(D.Jacobi()); b = poisson.rhs(S.n, 0)
    for name, fl in (("small", 0), ("general", D._lib.NO_SMALL)):
        S.solve(b, flags=fl); t = [S.solve(b, flags=fl, want_history=False) for _ in range(5)]
        r = min(t, key=lambda r: r.seconds)
        print(f"poisson{dim}d n={n} N={S.n} {name:8s}: {r.iterations} its, {r.seconds*1e3:.3f} ms, {r.seconds/r.iterations*1e6:.2f} us/it, {r.iterations/r.seconds:.0f} it/s")
# a batch of 256 systems of the reference's real size class (2.4k-5.5k rows), one launch
systems, rhs = [], []
for i in range(256):
    S = poisson.poisson_system(2, 49 + (i % 4)); S.set_preconditioner(D.Jacobi()); systems.append(S); rhs.append(poisson.rhs(S.n, i))
solve_batch(systems, rhs)
torch.cuda.synchronize(); t0 = time.perf_counter(); out = solve_batch(systems, rhs); torch.cuda.synchronize(); dt = time.perf_counter() - t0
its = sum(o.iterations for o in out)
print(f"batch of 256 systems (N~2.4-2.7k) in one launch: {dt*1e3:.2f} ms total, {its} iterations, {its/dt:.0f} it/s aggregate, {256/dt:.0f} systems/s")
t0 = time.perf_counter(); out2 = [s.solve(b, want_history=False) for s, b in zip(systems, rhs)]; dt2 = time.perf_counter() - t0
print(f"same 256 systems one after another (small kernel each): {dt2*1e3:.2f} ms")
t0 = time.perf_counter(); out3 = [s.solve(b, want_history=False, flags=D._lib.NO_SMALL) for s, b in zip(systems, rhs)]; dt3 = time.perf_counter() - t0
print(f"same 256 systems one after another (general 3-kernel path): {dt3*1e3:.2f} ms")
assert all(a.iterations == b_.iterations for a, b_ in zip(out, out3))
