"""Preconditioner setup times (the `setups` column of the harness, test.py:130-135) on large systems."""
import time
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

for dim, n in [(2, 256), (2, 1024), (3, 100)]:
    s = poisson.poisson_system(dim, n, device="cuda:0")
    for name, pc in (("jacobi", D.Jacobi()), ("ic0 multiply", D.IC0("multiply")), ("ic0 solve", D.IC0("solve"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.set_preconditioner(pc)
        torch.cuda.synchronize()
        print(f"poisson{dim}d_{n} ({s.n} rows): {name:13s} {1e3 * (time.perf_counter() - t0):9.2f} ms")
