"""Workload for `rocprofv3 --kernel-trace --stats`: the PCG loop kernels on BASELINE config 4's 256^3 system (1.74 GB per SpMV,
far beyond the Infinity Cache) -- the HBM-bound durations behind roofline.hbm_bound_256cubed."""
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from deeppreconditioning_amd import poisson  # noqa: E402

torch.cuda.set_device(0)
s = poisson.poisson_system(3, 256)
s.set_preconditioner(D.Jacobi())
b = poisson.rhs(s.n, 0)
s.solve(b, max_iter=8, want_history=False)
s.solve(b, max_iter=96, want_history=False)
s.spmv_dot_bench(20)
torch.cuda.synchronize()
