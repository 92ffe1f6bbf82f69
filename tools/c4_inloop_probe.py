"""256^3 (BASELINE config 4's system) inside the PCG loop: us per update of a 48-update solve, K1 back to back beside it.
    DPCG_VEC_NT=0|1 python tools/c4_inloop_probe.py"""
import sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
s4 = poisson.poisson_system(3, 256)
s4.set_preconditioner(D.Jacobi())
b4 = poisson.rhs(s4.n, 0)
s4.solve(b4, max_iter=8, want_history=False)
us = []
for _ in range(5):
    r4 = s4.solve(b4, max_iter=48, want_history=False)
    us.append(r4.seconds / max(r4.iterations, 1) * 1e6)
print(f"256^3 in-loop us/update: median {np.median(us):.1f} min {min(us):.1f}; K1 back to back {s4.spmv_dot_bench(repeats=40) * 1e3:.1f} us", flush=True)
