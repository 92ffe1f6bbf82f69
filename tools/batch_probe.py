"""Where a 256-system small batch spends its time: Python marshalling vs the C call vs the kernel."""
import time
import numpy as np
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import batch as B
from oracle import oracle as O

mats = [O.poisson2d(49 + (i % 4)) for i in range(256)]
systems = [D.CsrSystem.from_any(m) for m in mats]
for s in systems:
    s.set_preconditioner(D.Jacobi())
rhs = [torch.from_numpy(O.rhs(m.shape[0], i)).cuda() for i, m in enumerate(mats)]
for _ in range(3):
    out = B.solve_batch(systems, rhs)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    out = B.solve_batch(systems, rhs)
    ts.append(time.perf_counter() - t0)
print(f"solve_batch wall: median {np.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f} ms; iterations {sum(r.iterations for r in out)}")
print("kernel-side seconds reported by the library (max over systems):", max(r.seconds for r in out) * 1e3, "ms")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    B.solve_batch(systems, rhs)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
